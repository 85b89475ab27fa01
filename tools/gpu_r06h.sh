#!/bin/bash
# round 6, visit h: randomised engine-vs-oracle runs and the learning acceptance run on the final kernels
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
TAG=${1:-r06h}
SECONDS=0
timeout 1500 python tests/fuzz_engine.py 606 250 cuda 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_fuzz_engine_250.txt
echo "fuzz wall ${SECONDS}s"; grep -c "^OK" gpurun_out/${TAG}_fuzz_engine_250.txt; grep "^FAIL\|failed" gpurun_out/${TAG}_fuzz_engine_250.txt | cut -c1-400 | head -12
grep -c "critic_hid': 128" gpurun_out/${TAG}_fuzz_engine_250.txt
SECONDS=0
timeout 1500 python tools/learning_acceptance.py --seeds 64 --out gpurun_out/learning_${TAG}.json 2>&1 | grep -v "^This is\|^{'n_agents\|amdgpu.ids" | tail -12
echo "learning wall ${SECONDS}s"
