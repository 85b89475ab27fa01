#!/bin/bash
# round 6, visit c: parity evidence -- fp64 arbiter, steady-state block from identical state; PK tests
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
TAG=${1:-r06c}
nproc
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_sharded_engine_gpu.py -m gpu -q -k "pk or sharded" --maxfail=20 -p no:cacheprovider -rP > gpurun_out/${TAG}_tests.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed|packed-operand fit" gpurun_out/${TAG}_tests.log | tail -20
SECONDS=0
timeout 1500 python tools/diag_cfg4_fp64_arbiter.py 0 4 2> gpurun_out/${TAG}_arbiter.err | grep -v amdgpu.ids > gpurun_out/${TAG}_cfg4_fp64_arbiter.txt
echo "arbiter wall ${SECONDS}s"; cat gpurun_out/${TAG}_cfg4_fp64_arbiter.txt; tail -3 gpurun_out/${TAG}_arbiter.err
SECONDS=0
timeout 1800 python tools/diag_cfg4_steady.py 2 2> gpurun_out/${TAG}_steady.err | grep -v amdgpu.ids > gpurun_out/${TAG}_cfg4_steady_state_parity.txt
echo "steady wall ${SECONDS}s"; cat gpurun_out/${TAG}_cfg4_steady_state_parity.txt; tail -3 gpurun_out/${TAG}_steady.err
