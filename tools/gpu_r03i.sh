#!/bin/bash
# round 3, visit I: the two-piece f16 operand form (RCMARL_LAT_F16) -- kernel tests in every form, A/B of the GEMMs, whole suite, bench
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "lattice or mid_fit" 2>&1 | tail -15
for cfg in "0 -" "3 -" "3 0" "3 1" "1 -"; do
  set -- $cfg
  export RCMARL_LAT_F16=$1
  if [ "$2" = "-" ]; then unset RCMARL_LAT_W8; else export RCMARL_LAT_W8=$2; fi
  echo "#### RCMARL_LAT_F16=$1 RCMARL_LAT_W8=$2"
  timeout 300 python tools/kbench.py lattice 2>&1 | grep -v "^==" | tee -a gpurun_out/r03i_kbench_lattice.txt
done
unset RCMARL_LAT_F16 RCMARL_LAT_W8
bash tools/gpu_visit.sh r03i tests bench
