#!/bin/bash
# wide-critic GPU checks + cfg5 single-GPU bench (run through gpurun)
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_engine_gpu.py tests/test_kernels_gpu.py -q -p no:cacheprovider -x -k wide 2>&1 | tail -6
timeout 900 python bench.py --workload cfg5_1gpu --steps 1 --warmup 1 > gpurun_out/bench_cfg5_1gpu.json 2> gpurun_out/bench_cfg5_1gpu.err
echo "bench rc=$?"; tail -3 gpurun_out/bench_cfg5_1gpu.err; python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/bench_cfg5_1gpu.json").read().strip().splitlines()[-1])
    print({k: d[k] for k in ("value", "ms_per_step")}, d["phase_seconds_per_block"])
    for k, v in d["kernels"].items():
        print("%-32s %5d launches %10.1f ms  avg %10.1f us" % (k, v["launches"], v["total_ms"], v["avg_us"]))
    print(d["roofline"])
    print(d.get("cpu_baseline"))
except Exception as e:
    print("no bench line:", e)
PY
