#!/bin/bash
# round 6, visit i: the evidence run of the final build -- fuzz A/B of the K2 form, rocprofv3 kernel stats + PMC traffic of both headline
# workloads, the GPU suite, the bench line as the driver runs it
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
TAG=${1:-r06i}
echo "== fuzz, K2 on the vector ALUs (the same 250 draws)"
RCMARL_K2_MX=0 timeout 900 python tests/fuzz_engine.py 606 250 cuda 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_fuzz_engine_250_k2_fp32.txt
grep "^FAIL\|failed" gpurun_out/${TAG}_fuzz_engine_250_k2_fp32.txt | cut -c1-300
echo "== fuzz, default, another 250 draws"
timeout 900 python tests/fuzz_engine.py 707 250 cuda 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_fuzz_engine_250_seed707.txt
grep "^FAIL\|failed" gpurun_out/${TAG}_fuzz_engine_250_seed707.txt | cut -c1-300
echo "== rocprof + pmc cfg4_shard"
bash tools/gpu_visit.sh ${TAG} prof:cfg4_shard pmc:cfg4_shard 2>&1 | tail -22 | cut -c1-170
echo "== pmc cfg5_1gpu"
bash tools/gpu_visit.sh ${TAG} pmc:cfg5_1gpu 2>&1 | tail -12 | cut -c1-170
echo "== full GPU suite"
SECONDS=0
timeout 2400 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider --durations=6 -rP > gpurun_out/${TAG}_test_gpu.log 2>&1
echo "suite wall ${SECONDS}s"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_test_gpu.log | tail -20
grep -E "^E  " gpurun_out/${TAG}_test_gpu.log | head -20
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
echo "== bench (no flags: the driver's default)"
SECONDS=0
timeout 1200 python bench.py 2> gpurun_out/${TAG}_bench_default.err > gpurun_out/${TAG}_bench_default_flags.json
echo "bench.py wall: ${SECONDS}s"
echo "== bench (20 steps)"
SECONDS=0
timeout 1200 python bench.py --steps 20 --warmup 5 2> gpurun_out/${TAG}_bench.err > gpurun_out/${TAG}_bench_cfg4_shard.json
echo "bench.py wall: ${SECONDS}s"
python - <<PY
import json
for f in ('gpurun_out/${TAG}_bench_default_flags.json', 'gpurun_out/${TAG}_bench_cfg4_shard.json'):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1])
    print({k:d.get(k) for k in ('value','ms_per_step','ms_per_step_exact','exact_steps','steps','n_gpus')})
    print('  ', d.get('summary_ms_per_step'))
    print('  roofline', {k:d['roofline'].get(k) for k in ('kernel','achieved','frac','avg_us','traffic')})
PY
