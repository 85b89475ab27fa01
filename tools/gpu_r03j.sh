#!/bin/bash
# round 3, visit J: f16 operand form with saturation (MODE.FP16_OVFL) -- kernel tests, divergence diagnostic, suite, bench, profile
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "lattice or mid_fit" 2>&1 | tail -8
MODES=3 BLOCKS=5 timeout 300 python tools/diag_f16.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03j_diag_f16.txt
bash tools/gpu_visit.sh r03j tests bench prof:cfg4_shard
