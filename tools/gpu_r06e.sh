#!/bin/bash
# round 6, visit e: wide path after the cached-activation shortcuts + |phi|^2 parts; cfg5 block, rocprof; fp64 arbiter at the steady state
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
TAG=${1:-r06e}
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py tests/test_engine_baseline_shapes_gpu.py tests/test_sharded_engine_gpu.py tests/test_engine_round3_parity_gpu.py tests/test_rccl_one_rank_gpu.py -m gpu -q -k "pk or wide or packed or sharded or d66 or cfg5 or copy3d or multi or rccl" --maxfail=20 -p no:cacheprovider -rP > gpurun_out/${TAG}_tests.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_tests.log | tail -20
grep -E "^E  " gpurun_out/${TAG}_tests.log | head -20
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -4
echo "== bench cfg5_1gpu"
timeout 900 python bench.py --workload cfg5_1gpu --steps 3 --warmup 1 --no-extra --no-cpu-baseline --kernel-timing-steps 1 2> gpurun_out/${TAG}_bench_cfg5.err > gpurun_out/${TAG}_bench_cfg5_1gpu.json
tail -3 gpurun_out/${TAG}_bench_cfg5.err
python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/${TAG}_bench_cfg5_1gpu.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step','n_gpus')}, d['phase_seconds_per_block'])
for k,v in list(d['kernels'].items())[:18]: print('  ',k, v)
print(d.get('roofline'))
PY
echo "== rocprof cfg5_1gpu"
bash tools/gpu_visit.sh ${TAG} prof:cfg5_1gpu 2>&1 | tail -14 | cut -c1-170
echo "== arbiter at the steady state"
SECONDS=0
timeout 2400 python tools/diag_cfg4_fp64_arbiter.py 2 2 2> gpurun_out/${TAG}_arbiter.err | grep -v amdgpu.ids > gpurun_out/${TAG}_cfg4_fp64_arbiter_steady.txt
echo "arbiter wall ${SECONDS}s"; cat gpurun_out/${TAG}_cfg4_fp64_arbiter_steady.txt; tail -3 gpurun_out/${TAG}_arbiter.err
