#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
for rank in 1 2; do
  for mode in 0 1 2; do
  SEED0=$((1000 + 16 * rank)) MODES=$mode BLOCKS=8 timeout 120 python tools/diag_f16.py 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a gpurun_out/r03ah_seed_shards_modes.txt
  done
  SEED0=$((1000 + 16 * rank)) MODES=3 RCMARL_MIDFIT=5 BLOCKS=8 timeout 120 python tools/diag_f16.py 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/^/MIDFIT=5 /' | tee -a gpurun_out/r03ah_seed_shards_modes.txt
done
