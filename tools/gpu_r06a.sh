#!/bin/bash
# round 6, visit a: the packed-operand dense path -- parity, kernel timings, cfg5 block
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
TAG=${1:-r06a}
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py tests/test_engine_baseline_shapes_gpu.py tests/test_sharded_engine_gpu.py -m gpu -q -k "pk or wide or packed or sharded or d66 or 66" --maxfail=20 -p no:cacheprovider -rP > gpurun_out/${TAG}_tests.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed|packed-operand fit|parity\]" gpurun_out/${TAG}_tests.log | tail -40
grep -E "^E  " gpurun_out/${TAG}_tests.log | head -30
echo "== knife"; timeout 300 python tools/diag_pk_knife.py 2 16 777 3 7 9 128 3 0.02 4 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_knife.txt | grep -B1 -A6 "<--" | head -40
echo "== kbench pk"; timeout 600 python tools/kbench.py pk 2>&1 | tail -16 | tee gpurun_out/${TAG}_kbench_pk.txt
echo "== kbench wide"; timeout 300 python tools/kbench.py wide 2>&1 | tail -20 | tee gpurun_out/${TAG}_kbench_wide.txt
echo "== bench cfg5_1gpu"
timeout 900 python bench.py --workload cfg5_1gpu --steps 2 --warmup 1 --no-extra --no-cpu-baseline --kernel-timing-steps 1 2> gpurun_out/${TAG}_bench_cfg5.err > gpurun_out/${TAG}_bench_cfg5_1gpu.json
tail -3 gpurun_out/${TAG}_bench_cfg5.err
python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/${TAG}_bench_cfg5_1gpu.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step','n_gpus')}, d['phase_seconds_per_block'])
for k,v in list(d['kernels'].items())[:16]: print('  ',k, v)
PY
