#!/bin/bash
# HBM-side traffic (FETCH_SIZE doubled + WRITE_SIZE, MI355X_MICROARCH.md HBM section) of EVERY kernel a tools/kbench.py run launches,
# from two separate rocprofv3 --pmc passes (no tracing).   gpurun --timeout 600 -- 'bash tools/gpu_pmc_kbench.sh fwdmid r04p'
R=${GRAFT_REPO_ROOT:-/root/repo}
WHAT=${1:-fwdmid}; TAG=${2:-r04}
mkdir -p $R/gpurun_out/pmckb
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmckb -o ${WHAT}_$c -- python $R/tools/kbench.py $WHAT > $R/gpurun_out/pmckb/${WHAT}_$c.log 2>&1
done
python - <<PY
import csv, glob, collections, json, re
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for f in glob.glob('$R/gpurun_out/pmckb/${WHAT}_*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = re.sub(r'\(.*', '', r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', ''))
        if k.startswith('at::') or 'rocclr' in k: continue
        tot[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k][r['Counter_Name']] += 1
out = {}
for k in tot:
    f = tot[k].get('FETCH_SIZE', 0.0) / max(cnt[k].get('FETCH_SIZE', 1), 1); w = tot[k].get('WRITE_SIZE', 0.0) / max(cnt[k].get('WRITE_SIZE', 1), 1)
    out[k] = {"traffic_MB_per_launch": round((2.0 * f + w) * 1024.0 / 1e6, 1), "fetch_MB": round(2.0 * f * 1024 / 1e6, 1), "write_MB": round(w * 1024 / 1e6, 1),
              "launches_seen": int(cnt[k].get('FETCH_SIZE', 0))}
json.dump({"command": "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} -- python tools/kbench.py $WHAT", "env": {k: v for k, v in __import__('os').environ.items() if k.startswith('KB_')},
           "note": "averaged over every launch of the kernel in the run (both input widths unless KB_WIDTHS restricts them)", "kernels": out},
          open('$R/gpurun_out/${TAG}_pmc_kbench_$WHAT.json', 'w'), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["traffic_MB_per_launch"])[:14]: print("%-56s %s" % (k[:56], v))
PY
