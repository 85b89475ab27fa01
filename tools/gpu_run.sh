#!/bin/bash
# One GPU visit: parity tests, smoke, bench (+ optional rocprof kernel trace).  Run via:
#   gpurun --timeout 1500 -- 'bash tools/gpu_run.sh [prof]'
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
echo "== host: $(nproc) cpus"
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > gpurun_out/test_gpu.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/test_gpu.log | tail -30
grep -E "^E  " gpurun_out/test_gpu.log | head -40
echo "== smoke"
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -4 | tee gpurun_out/smoke.log
echo "== bench"
timeout 900 python bench.py 2> gpurun_out/bench.err > gpurun_out/bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench.json'))
print({k:d[k] for k in ('value','ms_per_step','consensus_updates_per_s','consensus_updates_per_s_phase2_only','phase_seconds_per_block')})
for k,v in list(d['kernels'].items())[:8]: print(k, v)
print('roofline', d['roofline']); print('roofline_consensus', d['roofline_consensus']); print('roofline_gemm', d.get('roofline_gemm')); print('cpu', d.get('cpu_baseline',{}).get('value'), d.get('speedup_vs_cpu_port'))
PY
tail -3 gpurun_out/bench.err
echo "== bench target workload"
timeout 600 python bench.py --steps 2 --warmup 2 --workload target_N256_H1 --no-cpu-baseline 2> gpurun_out/bench_target.err > gpurun_out/bench_target.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_target.json'))
print({k:d[k] for k in ('value','ms_per_step','consensus_updates_per_s_phase2_only')}); print('roofline_consensus', d['roofline_consensus'])
PY
if [ "${1:-}" = "prof" ]; then
echo "== bench without per-launch events (host/event overhead check)"
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('no-events ms_per_step', d['ms_per_step'], 'value', d['value'])"
echo "== rocprof"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o r01 -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err
tail -2 $R/gpurun_out/prof.err
ls $R/gpurun_out/prof | head
f=$(find $R/gpurun_out/prof -name '*kernel_stats*' | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-220
cd $R; bash tools/gpu_pmc_bench.sh cfg4_shard ${2:-r01}
fi
