#!/bin/bash
# candidate fast_lr for the 256-agent workloads: finite (and not blown up) on all 8 seed shards after 13 blocks?
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
for lr in 0.001 0.0015; do
for rank in 0 1 2 3 4 5 6 7; do
  FAST_LR=$lr SEED0=$((1000 + 16 * rank)) MODES=3 BLOCKS=13 timeout 120 python tools/diag_f16.py 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/lr=$lr /" | tee -a gpurun_out/r03ai_fast_lr_seed_shards.txt
done
done
