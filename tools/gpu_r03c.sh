#!/bin/bash
# Round-3 visit C: the bf16 matrix-core mid kernel (k_mid_fit_v7) against v5 and a 3-waves-per-SIMD build; SQ counters; tests; bench.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
TAG=${1:-r03c}
V=resilient-consensus-based-marl_amd/lib/variants
echo "== mid A/B"; RCMARL_KBENCH_LIB_B=$(ls $V/lib*.so 2>/dev/null | tr '\n' ',') timeout 300 python tools/kbench.py mid_ab 2>&1 | tail -40
echo "== lattice chain"; timeout 300 python tools/kbench.py lattice 2>&1 | tail -14
echo "== tests"; bash tools/gpu_visit.sh $TAG tests
grep -F "[parity]" gpurun_out/${TAG}_test_gpu.log | sort -u | head -40
echo "== bench"; bash tools/gpu_visit.sh $TAG bench
echo "== SQ counters of k_mid_fit_v7"; bash tools/gpu_pmc_kernel.sh lattice k_mid_fit_v7 $TAG 2>&1 | tail -30
