#!/bin/bash
# round 6, visit g: K2 on the matrix core, chunk-walking form (A/B + counters), GPU suite timing, headline bench
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
TAG=${1:-r06g}
echo "== kbench mid"; timeout 300 python tools/kbench.py mid 2>&1 | grep -v amdgpu.ids | tail -12 | tee gpurun_out/${TAG}_kbench_mid.txt
echo "== SQ counters of the K2 kernel at (18, 8)"
RCMARL_KBENCH_ONLY=18 bash tools/gpu_pmc_kernel.sh mid k_consensus_head_mx ${TAG} 2>&1 | grep -E "^(SQ_INSTS|SQ_VALU_MFMA|SQ_LDS_BANK|SQ_LDS_IDX|SQ_WAVES|GRBM)" | sort -u
echo "== full GPU suite"
SECONDS=0
timeout 2400 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider --durations=8 -rP > gpurun_out/${TAG}_test_gpu.log 2>&1
echo "suite wall ${SECONDS}s"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_test_gpu.log | tail -30
grep -E "^E  " gpurun_out/${TAG}_test_gpu.log | head -30
grep -A10 "slowest" gpurun_out/${TAG}_test_gpu.log | head -12
grep -E "steady state" gpurun_out/${TAG}_test_gpu.log | head -8
echo "== bench (driver form, 20 steps)"
SECONDS=0
timeout 1200 python bench.py --steps 20 --warmup 5 2> gpurun_out/${TAG}_bench.err > gpurun_out/${TAG}_bench_cfg4_shard.json
echo "bench.py wall: ${SECONDS}s"; tail -2 gpurun_out/${TAG}_bench.err
python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/${TAG}_bench_cfg4_shard.json') if l.startswith('{')][-1])
print({k:d.get(k) for k in ('value','ms_per_step','ms_per_step_exact','exact_steps','n_gpus')}, d['phase_seconds_per_block'])
for k,v in list(d['kernels'].items())[:8]: print('  ',k, v)
for r in ('roofline','roofline_consensus','roofline_gemm','roofline_mid','roofline_consensus_target'):
    if d.get(r): print('  ',r, {k:d[r].get(k) for k in ('kernel','achieved','frac','avg_us')})
print('  ', d.get('summary_ms_per_step'))
PY
