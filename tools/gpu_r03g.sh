#!/bin/bash
# Round-3 visit G: the build with one gradient record per mid-fit workgroup: GPU suite + default bench line + kernel stats.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; TAG=${1:-r03g}
bash tools/gpu_visit.sh $TAG tests bench prof:cfg4_shard
