#!/bin/bash
# round 3, visit V: the adversaries' mini-batch fit on the f16 matrix core (k_minibatch_mx) -- tests, A/B, engine tests with adversaries, bench
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "minibatch" 2>&1 | tail -6
for m in 1 0; do echo "#### RCMARL_MB_MX=$m"; RCMARL_MB_MX=$m timeout 200 python tools/kbench.py minibatch 2>&1 | grep -v "amdgpu.ids\|^==" | tee -a gpurun_out/r03v_kbench_minibatch.txt; done
timeout 600 python -m pytest tests/test_engine_gpu.py tests/test_dropin_gpu.py -m gpu -q -p no:cacheprovider -k "advers or malicious or greedy or Malicious or bit_identical or dropin" -rP 2>&1 | grep -E "passed|failed|parity|Error" | tail -20
bash tools/gpu_visit.sh r03v bench
