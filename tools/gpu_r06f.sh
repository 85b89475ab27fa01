#!/bin/bash
# round 6, visit f: K2 on the matrix core (A/B), the headline bench, the arbiter at the steady state with the oracle spread over the host's cores
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
TAG=${1:-r06f}
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -m gpu -q -k "consensus_head or engine_matches or lattice_path or baseline" --maxfail=20 -p no:cacheprovider -rP > gpurun_out/${TAG}_tests.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_tests.log | tail -8
grep -E "^E  " gpurun_out/${TAG}_tests.log | head -20
echo "== kbench mid"; timeout 300 python tools/kbench.py mid 2>&1 | grep -v amdgpu.ids | tail -12 | tee gpurun_out/${TAG}_kbench_mid.txt
echo "== SQ counters of the K2 kernel at (18, 8)"
RCMARL_KBENCH_ONLY=18 bash tools/gpu_pmc_kernel.sh mid k_consensus_head_mx ${TAG} 2>&1 | tail -30
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
print('  target', (d.get('extra') or {}).get('target_N256_H1', {}).get('speedup_vs_cpu_port'), 'cpu', d.get('cpu_baseline',{}).get('value'), d.get('speedup_vs_cpu_port'))
PY
echo "== arbiter at the steady state (oracle spread over the cores)"
SECONDS=0
timeout 1200 python tools/diag_cfg4_fp64_arbiter.py 2 2 2> gpurun_out/${TAG}_arbiter.err | grep -v amdgpu.ids > gpurun_out/${TAG}_cfg4_fp64_arbiter_steady.txt
echo "arbiter wall ${SECONDS}s"; cat gpurun_out/${TAG}_cfg4_fp64_arbiter_steady.txt; tail -3 gpurun_out/${TAG}_arbiter.err
