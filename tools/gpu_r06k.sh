#!/bin/bash
# round 6, visit k: K2 without the low x low pass (A/B against the four-pass variant), cost of the K = 8 matrix instruction
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
TAG=${1:-r06k}
hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_k8_probe tools/micro/mfma_k8_probe.hip 2>/dev/null && /tmp/mfma_k8_probe | tee gpurun_out/${TAG}_mfma_k8_probe.txt
for rep in 1 2; do
echo "== kbench mid (product build: three passes)"; RCMARL_KBENCH_ONLY_K2=1 timeout 300 python tools/kbench.py mid 2>&1 | grep cons_head | tee -a gpurun_out/${TAG}_kbench_mid.txt
echo "== kbench mid (variant: four passes)"
RCMARL_KBENCH_LIB=resilient-consensus-based-marl_amd/lib/variants/libk2ll.so timeout 300 python tools/kbench.py mid 2>&1 | grep cons_head | tee -a gpurun_out/${TAG}_kbench_mid_4pass.txt
done
echo "== kernel tests"
timeout 1200 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -k "consensus or head" 2>&1 | tail -4
