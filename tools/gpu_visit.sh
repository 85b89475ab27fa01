#!/bin/bash
# One GPU visit: parity tests, the default bench line (with extras), optional learning acceptance, rocprof of a workload.
#   gpurun --timeout 1500 -- 'bash tools/gpu_visit.sh TAG [tests] [bench] [learn] [prof:WORKLOAD] [pmc:WORKLOAD]'
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
TAG=$1; shift
for what in "$@"; do
case $what in
tests)
  timeout 1200 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider --durations=10 -rP > gpurun_out/${TAG}_test_gpu.log 2>&1
  grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_test_gpu.log | tail -30
  grep -E "^E  " gpurun_out/${TAG}_test_gpu.log | head -40
  grep -A12 "slowest" gpurun_out/${TAG}_test_gpu.log | head -14 ;;
bench)
  SECONDS=0
  timeout 900 python bench.py --steps 10 --warmup 3 2> gpurun_out/${TAG}_bench.err > gpurun_out/${TAG}_bench_cfg4_shard.json
  echo "bench.py wall: ${SECONDS}s"; tail -3 gpurun_out/${TAG}_bench.err
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/${TAG}_bench_cfg4_shard.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step','n_gpus')}, d['phase_seconds_per_block'])
for k,v in list(d['kernels'].items())[:8]: print('  ',k, v)
for r in ('roofline','roofline_consensus','roofline_gemm','roofline_mid','roofline_consensus_target'):
    if d.get(r): print('  ',r, {k:d[r].get(k) for k in ('kernel','achieved','frac','avg_us')})
for k,v in d.get('extra',{}).items(): print('  extra',k, {q:v.get(q) for q in ('ms_per_step','agent_steps_per_s','weights_finite','epochs_replayed_from_hipgraph','error')})
print('  cpu', d.get('cpu_baseline',{}).get('value'), d.get('speedup_vs_cpu_port'))
PY
  ;;
learn)
  timeout 1200 python tools/learning_acceptance.py --seeds 64 --out gpurun_out/${TAG}_learning.json 2>&1 | grep -v "^This is\|^{'n_agents" | tail -12 ;;
prof:*)
  W=${what#prof:}
  ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_$W -o $W -- python $R/bench.py --steps 3 --warmup 2 --workload $W --no-cpu-baseline --no-extra > $R/gpurun_out/${TAG}_prof_$W.log 2>&1 )
  f=$(ls gpurun_out/prof_${TAG}_$W/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp $f gpurun_out/${TAG}_kernel_stats_$W.csv && head -12 $f | cut -c1-160 ;;
kbench:*)
  timeout 300 python tools/kbench.py ${what#kbench:} 2>&1 | tail -30 ;;
env:*)
  export ${what#env:} ;;
pmc:*)
  W=${what#pmc:}
  bash tools/gpu_pmc_bench.sh $W $TAG | tail -12 ;;
esac
done
