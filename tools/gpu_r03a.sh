#!/bin/bash
# Round-3 visit A: cycle-level pipe-overlap micro, int8/bf16 matrix-core rates with clocks, GPU parity suite (with the
# one-rank RCCL test), every collective + cfg5_shard under a one-rank RCCL group, effective clocks of the hot kernels.
#   gpurun --timeout 1500 -- 'bash tools/gpu_r03a.sh r03a'
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
TAG=${1:-r03a}
echo "== micro: pipe_overlap_cycles"; timeout 120 tools/micro/pipe_overlap_cycles > gpurun_out/${TAG}_pipe_overlap_cycles.txt 2>&1; tail -70 gpurun_out/${TAG}_pipe_overlap_cycles.txt | cut -c1-330
echo "== micro: mfma_peak (bf16 / int8 + clocks)"; timeout 120 tools/micro/mfma_peak > gpurun_out/${TAG}_mfma_peak.txt 2>&1; tail -20 gpurun_out/${TAG}_mfma_peak.txt
echo "== tests"; bash tools/gpu_visit.sh $TAG tests
bjson() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    print('   ', d['config']['workload'], 'ms/block', round(d['ms_per_step'], 2), 'agent-steps/s', round(d['value']), 'comm', d.get('comm'), 'finite', d['config']['weights_finite'], d['phase_seconds_per_block'])
    for k, v in list(d.get('kernels', {}).items())[:6]: print('      ', k, v)
except Exception as e:
    print('    no bench line:', e); print(open(sys.argv[1]).read()[-1500:])
PY
}
echo "== one-rank RCCL group: default workload (C1 all-reduce on a device tensor)"
RCMARL_BENCH_FORCE_PG=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 3 --warmup 1 --no-extra --no-cpu-baseline > gpurun_out/${TAG}_bench_rccl1_cfg4_shard.json 2> gpurun_out/${TAG}_bench_rccl1_cfg4_shard.err; bjson gpurun_out/${TAG}_bench_rccl1_cfg4_shard.json
echo "== one-rank RCCL group: cfg5_shard (C2 all-to-all + all-gathers through RCCL) vs cfg5_1gpu"
RCMARL_BENCH_FORCE_PG=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 1 --workload cfg5_shard --steps 1 --warmup 1 --no-extra --no-cpu-baseline > gpurun_out/${TAG}_bench_rccl1_cfg5_shard.json 2> gpurun_out/${TAG}_bench_rccl1_cfg5_shard.err; bjson gpurun_out/${TAG}_bench_rccl1_cfg5_shard.json
timeout 600 python bench.py --workload cfg5_1gpu --steps 1 --warmup 1 --no-extra --no-cpu-baseline > gpurun_out/${TAG}_bench_cfg5_1gpu.json 2> gpurun_out/${TAG}_bench_cfg5_1gpu.err; bjson gpurun_out/${TAG}_bench_cfg5_1gpu.json
echo "== single-instance (S=1) and H=0 workloads"
for w in cfg0_H0_batched cfg0_H0_single cfg2_single cfg3_single; do
  timeout 300 python bench.py --workload $w --steps 3 --warmup 1 --no-extra --no-cpu-baseline > gpurun_out/${TAG}_bench_$w.json 2> gpurun_out/${TAG}_bench_$w.err; bjson gpurun_out/${TAG}_bench_$w.json
done
echo "== effective clocks of the hot kernels (GRBM_GUI_ACTIVE pass + kernel-trace pass of the same command)"
bash tools/gpu_clocks.sh cfg4_shard $TAG
