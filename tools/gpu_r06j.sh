#!/bin/bash
# round 6, visit j: K2 on the matrix core, third form (biases in the tables, scaled accumulators, no register copies): A/B, counters, tests
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
TAG=${1:-r06j}
echo "== kbench mid (product build: 3 wavefronts per SIMD)"; timeout 300 python tools/kbench.py mid 2>&1 | grep -v amdgpu.ids | tail -8 | tee gpurun_out/${TAG}_kbench_mid.txt
echo "== kbench mid (variant: register budget of 4 wavefronts per SIMD, 20 B of scratch)"
RCMARL_KBENCH_LIB=resilient-consensus-based-marl_amd/lib/variants/libk2w4.so timeout 300 python tools/kbench.py mid 2>&1 | grep cons_head | tee gpurun_out/${TAG}_kbench_mid_w4.txt
echo "== kernel tests"
timeout 1200 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -k "consensus or head" 2>&1 | tail -8
echo "== SQ counters of the K2 kernel at (18, 8)"
RCMARL_KBENCH_ONLY=18 bash tools/gpu_pmc_kernel.sh mid k_consensus_head_mx ${TAG} 2>&1 | grep -E "^(SQ_INSTS|SQ_VALU_MFMA|SQ_LDS_BANK|SQ_ACTIVE_INST_VALU|SQ_WAVES|GRBM|SQ_BUSY_CY)" | sort -u
