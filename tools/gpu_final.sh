#!/bin/bash
# round-end validation: parity tests, default bench line (+extras), rocprof stats + PMC of the default workload
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; TAG=${1:-r02z}
bash tools/gpu_visit.sh $TAG tests bench prof:cfg4_shard pmc:cfg4_shard
for v in 2 5; do
  echo "== RCMARL_MIDFIT=$v cfg1_batched / cfg3"
  for w in cfg1_batched cfg3; do
    RCMARL_MIDFIT=$v timeout 300 python bench.py --steps 3 --warmup 1 --workload $w --no-extra --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ', d['config']['workload'], round(d['ms_per_step'],1), 'ms/block', d['config']['weights_finite'])"
  done
done
