#!/bin/bash
# Effective shader clock per kernel of a bench workload: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / average duration.
# Two passes of the same command (PMC alone, then kernel-trace alone, as gpurun requires).
#   bash tools/gpu_clocks.sh [workload] [tag]   -> gpurun_out/<tag>_clocks_<workload>.json
R=${GRAFT_REPO_ROOT:-/root/repo}
W=${1:-cfg4_shard}; TAG=${2:-r03}
mkdir -p $R/gpurun_out/clk
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 1 --warmup 1 --workload $W --no-cpu-baseline --no-kernel-timing --no-extra"
timeout 400 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/clk -o ${W}_pmc -- $CMD > $R/gpurun_out/clk/${W}_pmc.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/clk -o ${W}_trace -- $CMD > $R/gpurun_out/clk/${W}_trace.log 2>&1
python - <<PY
import csv, glob, collections, json, re
def key(n):
    m = re.search(r'(k_\w+|at::native::\w+|__amd_\w+)', n)
    return m.group(1) if m else n[:40]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for f in glob.glob('$R/gpurun_out/clk/${W}_pmc*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = key(r['Kernel_Name'])
        tot[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k][r['Counter_Name']] += 1
dur = {}
for f in glob.glob('$R/gpurun_out/clk/${W}_trace*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        dur[key(r['Name'])] = (float(r['AverageNs']), int(r['Calls']), float(r['Percentage']))
out = {}
for k, (ns, calls, pct) in sorted(dur.items(), key=lambda kv: -kv[1][2])[:14]:
    if k not in tot: continue
    gui = tot[k]['GRBM_GUI_ACTIVE'] / max(cnt[k]['GRBM_GUI_ACTIVE'], 1)
    sqb = tot[k]['SQ_BUSY_CYCLES'] / max(cnt[k]['SQ_BUSY_CYCLES'], 1)
    out[k] = {"avg_us": ns / 1e3, "calls": calls, "pct_of_gpu_time": pct, "GRBM_GUI_ACTIVE_per_launch": gui, "SQ_BUSY_CYCLES_per_launch": sqb,
              "effective_GHz_gui_over_8xcd": gui / 8 / ns, "effective_GHz_sqbusy_over_32se": sqb / 32 / ns}
    print("%-60s %9.1f us  %5.1f %%  GUI/8/t = %.2f GHz   SQ_BUSY/32/t = %.2f GHz" % (k, ns / 1e3, pct, gui / 8 / ns, sqb / 32 / ns))
json.dump({"workload": "$W", "note": "effective clock = GRBM_GUI_ACTIVE (all XCDs) / 8 / average kernel duration of a separate kernel-trace pass of the same command (MI355X_MICROARCH.md, DVFS give-back)", "kernels": out}, open('$R/gpurun_out/${TAG}_clocks_$W.json', 'w'), indent=1)
PY
