#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
run() { echo "== $*"; env "$@" timeout 300 python bench.py --steps 1 --warmup 1 --no-extra --no-cpu-baseline --no-kernel-timing 2>&1 | tail -1 | cut -c1-200; }
run RCMARL_FIT_FUSED=0 X=1
run RCMARL_FIT_FUSED=1 X=1
