#!/bin/bash
# round 6, visit u: the final build -- GPU suite, fuzz, bench, rocprof
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
TAG=${1:-r06u}
echo "== full GPU suite"
SECONDS=0
timeout 2400 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider --durations=6 -rP > gpurun_out/${TAG}_test_gpu.log 2>&1
echo "suite wall ${SECONDS}s"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_test_gpu.log | tail -20
grep -E "^E  " gpurun_out/${TAG}_test_gpu.log | head -20
echo "== fuzz, 250 draws (seed 808)"
timeout 900 python tests/fuzz_engine.py 808 250 cuda 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_fuzz_engine_250_seed808.txt
grep "^FAIL\|failed" gpurun_out/${TAG}_fuzz_engine_250_seed808.txt | cut -c1-400
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
echo "== bench (20 steps)"
timeout 1200 python bench.py --steps 20 --warmup 5 2> gpurun_out/${TAG}_bench.err > gpurun_out/${TAG}_bench_cfg4_shard.json
python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/${TAG}_bench_cfg4_shard.json') if l.startswith('{')][-1])
print({k:d.get(k) for k in ('value','ms_per_step','ms_per_step_exact','exact_steps','steps','n_gpus')})
print('  ', d.get('summary_ms_per_step'))
for k,v in list(d['kernels'].items())[:9]: print('  ',k, v)
PY
echo "== rocprof cfg4_shard"
bash tools/gpu_visit.sh ${TAG} prof:cfg4_shard 2>&1 | tail -12 | cut -c1-170
