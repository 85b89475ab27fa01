#!/bin/bash
# round 6, visit n: issue priority around the matrix groups of the mid kernels (A/B), fp64 arbiter at the steady state on the three-pass build
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
TAG=${1:-r06n}
V=resilient-consensus-based-marl_amd/lib/variants
echo "== mid_ab: product | s_setprio 1 | s_setprio 3 around the matrix groups"
RCMARL_KBENCH_LIB_B=$V/libmidprio1.so,$V/libmidprio3.so timeout 600 python tools/kbench.py mid_ab 2>&1 | grep -v "amdgpu.ids\|product-v5" | tee gpurun_out/${TAG}_mid_ab_setprio.txt
for lib in "" $V/libmidprio1.so $V/libmidprio3.so; do
  echo "== kbench mid K2, lib=${lib:-product}"
  RCMARL_KBENCH_LIB=$lib timeout 300 python tools/kbench.py mid 2>&1 | grep "cons_head.*MX=1" | tee -a gpurun_out/${TAG}_kbench_mid_setprio.txt
done
echo "== fp64 arbiter at the steady state, two seeds"
SECONDS=0
timeout 2400 python tools/diag_cfg4_fp64_arbiter.py 2 2 2> gpurun_out/${TAG}_arbiter.err | grep -v amdgpu.ids > gpurun_out/${TAG}_cfg4_fp64_arbiter_steady.txt
echo "arbiter wall ${SECONDS}s"; cat gpurun_out/${TAG}_cfg4_fp64_arbiter_steady.txt; tail -3 gpurun_out/${TAG}_arbiter.err
