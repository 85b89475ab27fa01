#!/bin/bash
# round 6, visit t: k_mid_value_mx (A/B, tests), the whole GPU suite and the bench line on the final build
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
TAG=${1:-r06t}
echo "== kernel tests"
timeout 1200 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -k "consensus or head or mid_value or sgd_fit" 2>&1 | tail -8
echo "== kbench mid"; timeout 300 python tools/kbench.py mid 2>&1 | grep -v amdgpu.ids | tail -10 | tee gpurun_out/${TAG}_kbench_mid.txt
echo "== full GPU suite"
SECONDS=0
timeout 2400 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider --durations=6 -rP > gpurun_out/${TAG}_test_gpu.log 2>&1
echo "suite wall ${SECONDS}s"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_test_gpu.log | tail -20
grep -E "^E  " gpurun_out/${TAG}_test_gpu.log | head -20
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
echo "== bench (20 steps)"
SECONDS=0
timeout 1200 python bench.py --steps 20 --warmup 5 2> gpurun_out/${TAG}_bench.err > gpurun_out/${TAG}_bench_cfg4_shard.json
echo "bench.py wall: ${SECONDS}s"
python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/${TAG}_bench_cfg4_shard.json') if l.startswith('{')][-1])
print({k:d.get(k) for k in ('value','ms_per_step','ms_per_step_exact','exact_steps','steps','n_gpus')})
print('  ', d.get('summary_ms_per_step'))
for k,v in list(d['kernels'].items())[:9]: print('  ',k, v)
PY
echo "== bench with RCMARL_MIDVALUE_MX=0 (same box)"
RCMARL_MIDVALUE_MX=0 timeout 1200 python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'])"
