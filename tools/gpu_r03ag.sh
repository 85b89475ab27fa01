#!/bin/bash
# the 8 seed shards the driver's --gpus 8 run uses (seeds 1000 + 16 rank + k): finite weights after 8 blocks on every shard?
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
for rank in 0 1 2 3 4 5 6 7; do
  SEED0=$((1000 + 16 * rank)) MODES=3 BLOCKS=8 timeout 120 python tools/diag_f16.py 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a gpurun_out/r03ag_seed_shards_finite.txt
done
