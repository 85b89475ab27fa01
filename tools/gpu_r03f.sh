#!/bin/bash
# Round-3 final visit: GPU suite, default bench line (with extras), rocprofv3 stats + PMC traffic of the default workload and of the
# north-star target shape, effective clocks, learning acceptance on the final build.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; TAG=${1:-r03f}
bash tools/gpu_visit.sh $TAG tests bench prof:cfg4_shard pmc:cfg4_shard prof:target_N256_H1 pmc:target_N256_H1
grep -F "[parity]" gpurun_out/${TAG}_test_gpu.log | sort -u > gpurun_out/${TAG}_parity_worst_cases.txt; wc -l gpurun_out/${TAG}_parity_worst_cases.txt
bash tools/gpu_visit.sh $TAG learn
