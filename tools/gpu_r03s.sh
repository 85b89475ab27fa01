#!/bin/bash
# round 3, visit S: k_mid_fit_v8 default + v5 fix-up -- blow-up diagnostic, mid A/B, suite, bench, profile
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
MODES=3 BLOCKS=4 timeout 250 python tools/diag_f16.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03s_diag_f16.txt
timeout 300 python tools/kbench.py mid_ab 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03s_mid_ab.txt
bash tools/gpu_visit.sh r03s tests bench prof:cfg4_shard
