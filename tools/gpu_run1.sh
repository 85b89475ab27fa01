#!/bin/bash
# One GPU visit: parity tests, smoke, bench, rocprof kernel trace.  Run via:
#   gpurun --timeout 1500 -- 'bash tools/gpu_run1.sh'
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
echo "== host: $(nproc) cpus; $(rocm-smi --showproductname 2>/dev/null | grep -m1 -i 'card series' || true)"
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider 2>&1 | tail -60 | tee gpurun_out/test_gpu.log
echo "== smoke"
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "== bench"
timeout 900 python bench.py --steps 2 --warmup 2 2> gpurun_out/bench.err | tee gpurun_out/bench.json
tail -5 gpurun_out/bench.err
echo "== bench target workload"
timeout 600 python bench.py --steps 2 --warmup 2 --workload target_N256_H1 --no-cpu-baseline 2> gpurun_out/bench_target.err | tee gpurun_out/bench_target.json
echo "== rocprof"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o r01 -- python $R/bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err
tail -3 $R/gpurun_out/prof.err
ls -la $R/gpurun_out/prof 2>/dev/null | head
find $R/gpurun_out/prof -name '*kernel_stats*' | head -2 | while read f; do echo "--- $f"; head -25 "$f"; done
