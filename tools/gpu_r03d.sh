#!/bin/bash
# Round-3 visit D: the bf16 matrix-core mid kernel with transposed-read reduction (k_mid_fit_v7) against v5; tests; bench; counters.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
TAG=${1:-r03d}
V=resilient-consensus-based-marl_amd/lib/variants
echo "== mid A/B"; RCMARL_KBENCH_LIB_B=$(ls $V/lib*.so 2>/dev/null | tr '\n' ',') timeout 300 python tools/kbench.py mid_ab 2>&1 | grep -v "round 0" | tail -30
echo "== tests (kernels only first)"; timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -p no:cacheprovider -k "sgd_fit or lattice" 2>&1 | tail -5
echo "== bench"; bash tools/gpu_visit.sh $TAG bench
echo "== tests"; bash tools/gpu_visit.sh $TAG tests
grep -F "[parity]" gpurun_out/${TAG}_test_gpu.log | sort -u | head -40
echo "== SQ counters of k_mid_fit_v7"; bash tools/gpu_pmc_kernel.sh lattice k_mid_fit_v7 $TAG 2>&1 | grep -v "rocprofv3\|^W2" | tail -28
