#!/bin/bash
# Round-3 visit E: the int8-limb forward prototype against the bf16x3 forward; full GPU suite with the tightened tolerances.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
TAG=${1:-r03e}
echo "== int8 forward prototype"; timeout 300 python tools/kbench.py i8 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_kbench_i8.txt | tail -20
echo "== tests"; bash tools/gpu_visit.sh $TAG tests
grep -F "[parity]" gpurun_out/${TAG}_test_gpu.log | sort -u | head -50
