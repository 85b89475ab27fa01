#!/bin/bash
# round 6, visit d: kernel evidence of the packed-operand path + the full suite + the bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
TAG=${1:-r06d}
echo "== kbench pk"; timeout 600 python tools/kbench.py pk 2>&1 | grep -v amdgpu.ids | tail -22 | tee gpurun_out/${TAG}_kbench_pk.txt
echo "== counters (fit-step kernels only)"
RCMARL_KBENCH_STEP_ONLY=1 bash tools/gpu_pmc_kbench_counters.sh pk ${TAG} "k_pk_|k_lat_" 2>&1 | tail -12
echo "== rocprof cfg5_1gpu"
bash tools/gpu_visit.sh ${TAG} prof:cfg5_1gpu 2>&1 | tail -14
echo "== full GPU suite"
SECONDS=0
timeout 2400 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider --durations=12 -rP > gpurun_out/${TAG}_test_gpu.log 2>&1
echo "suite wall ${SECONDS}s"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_test_gpu.log | tail -30
grep -E "^E  " gpurun_out/${TAG}_test_gpu.log | head -30
grep -A14 "slowest" gpurun_out/${TAG}_test_gpu.log | head -16
grep -E "steady state|packed-operand" gpurun_out/${TAG}_test_gpu.log | head -20
echo "== bench (driver form)"
SECONDS=0
timeout 1200 python bench.py --steps 10 --warmup 3 2> gpurun_out/${TAG}_bench.err > gpurun_out/${TAG}_bench_cfg4_shard.json
echo "bench.py wall: ${SECONDS}s"; tail -3 gpurun_out/${TAG}_bench.err
python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/${TAG}_bench_cfg4_shard.json') if l.startswith('{')][-1])
print({k:d.get(k) for k in ('value','ms_per_step','ms_per_step_exact','exact_steps','n_gpus')}, d['phase_seconds_per_block'])
for k,v in list(d['kernels'].items())[:8]: print('  ',k, v)
for r in ('roofline','roofline_consensus','roofline_gemm','roofline_mid','roofline_consensus_target'):
    if d.get(r): print('  ',r, {k:d[r].get(k) for k in ('kernel','achieved','frac','avg_us')})
for k,v in d.get('extra',{}).items(): print('  extra',k, {q:v.get(q) for q in ('ms_per_step','steps','agent_steps_per_s','weights_finite','epochs_replayed_from_hipgraph','error','speedup_vs_cpu_port')})
print('  cpu', d.get('cpu_baseline',{}).get('value'), d.get('speedup_vs_cpu_port'))
PY
