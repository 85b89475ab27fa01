#!/bin/bash
# GPU visit: bench only (default workload + optional others).  gpurun --timeout 900 -- 'bash tools/gpu_bench.sh [workloads...]'
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
for w in "${@:-cfg4_shard}"; do
timeout 600 python bench.py --steps 2 --warmup 2 --workload $w --no-cpu-baseline 2> gpurun_out/bench_$w.err > gpurun_out/bench_$w.json
python - <<PY
import json
d=json.load(open('gpurun_out/bench_$w.json'))
print('$w', {k:d[k] for k in ('value','ms_per_step','consensus_updates_per_s_phase2_only')}, d['phase_seconds_per_block'])
for k,v in list(d['kernels'].items())[:9]: print('  ',k, v)
print('  roofline', {k:d['roofline'][k] for k in ('kernel','achieved','frac','avg_us')}); print('  consensus', {k:d['roofline_consensus'][k] for k in ('achieved','frac','avg_us')})
PY
tail -2 gpurun_out/bench_$w.err
done
