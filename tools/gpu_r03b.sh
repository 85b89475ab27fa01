#!/bin/bash
# Round-3 visit B: tr_b16 probe, GPU parity suite of the pruned build (with [parity] worst cases), default bench line, rocprof.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
TAG=${1:-r03b}
echo "== probe"; timeout 60 tools/micro/tr_b16_probe > gpurun_out/${TAG}_tr_b16_probe.txt 2>&1; head -40 gpurun_out/${TAG}_tr_b16_probe.txt
echo "== tests"
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider --durations=12 -rP > gpurun_out/${TAG}_test_gpu.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_test_gpu.log | tail -30
grep -E "^E  " gpurun_out/${TAG}_test_gpu.log | head -40
grep -F "[parity]" gpurun_out/${TAG}_test_gpu.log | sort | uniq -c | sort -rn | head -60
grep -A14 "slowest" gpurun_out/${TAG}_test_gpu.log | head -16
bash tools/gpu_visit.sh $TAG bench prof:cfg4_shard
