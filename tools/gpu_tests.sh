#!/bin/bash
# GPU visit: parity tests only.   gpurun --timeout 900 -- 'bash tools/gpu_tests.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 800 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider --durations=8 > gpurun_out/test_gpu.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/test_gpu.log | tail -30
grep -E "^E  " gpurun_out/test_gpu.log | head -60
grep -A12 "slowest" gpurun_out/test_gpu.log | head -14
