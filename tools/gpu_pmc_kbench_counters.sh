#!/bin/bash
# SQ / L2 counters + durations of every kernel matching REGEX in a tools/kbench.py run (separate rocprofv3 --pmc passes, no tracing in
# them; one --kernel-trace --stats pass for the durations) -> gpurun_out/TAG_pmc_kbench_WHAT_counters.json with eff_clock_GHz,
# mfma_busy_frac (SQ_VALU_MFMA_BUSY_CYCLES per SIMD / cycles) and l2_hit_rate per kernel.
#   gpurun --timeout 900 -- 'bash tools/gpu_pmc_kbench_counters.sh pk r06 "k_pk_|k_lat_"'
R=${GRAFT_REPO_ROOT:-/root/repo}
WHAT=${1:-pk}; TAG=${2:-r06}; RE=${3:-k_pk_}
mkdir -p $R/gpurun_out/pmcc
i=0
for c in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmcc -o ${TAG}_c$i -- python $R/tools/kbench.py $WHAT > $R/gpurun_out/pmcc/${TAG}_c$i.log 2>&1 )
done
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pmcc -o ${TAG}_ct -- python $R/tools/kbench.py $WHAT > $R/gpurun_out/pmcc/${TAG}_ct.log 2>&1 )
python3 - <<PY
import csv, glob, collections, json, re
rx = re.compile(r'$RE')
def key(n):
    n = n.replace('void ', '').replace('(anonymous namespace)::', '')
    n = re.sub(r'\(.*', '', n)
    return n if rx.search(n) else None
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter); dur = {}
for f in glob.glob('$R/gpurun_out/pmcc/${TAG}_c[0-9]*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = key(r['Kernel_Name'])
        if k: tot[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k][r['Counter_Name']] += 1
for f in glob.glob('$R/gpurun_out/pmcc/${TAG}_ct*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        k = key(r['Name'])
        if k: dur[k] = (float(r['AverageNs']) / 1e3, int(r['Calls']))
out = {}
for k in tot:
    d = {c: tot[k][c] / cnt[k][c] for c in tot[k]}
    if k in dur:
        d['avg_us'], d['calls'] = dur[k]
        if 'GRBM_GUI_ACTIVE' in d:
            d['eff_clock_GHz'] = d['GRBM_GUI_ACTIVE'] / 8 / (d['avg_us'] * 1e3)
            if 'SQ_VALU_MFMA_BUSY_CYCLES' in d: d['mfma_busy_frac'] = d['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * d['eff_clock_GHz'] * 1e3 * d['avg_us'])
    if 'TCC_HIT_sum' in d: d['l2_hit_rate'] = d['TCC_HIT_sum'] / max(d['TCC_HIT_sum'] + d['TCC_MISS_sum'], 1)
    if 'FETCH_SIZE' in d: d['hbm_traffic_MB'] = (2.0 * d['FETCH_SIZE'] + d.get('WRITE_SIZE', 0.0)) * 1024 / 1e6
    out[k] = d
json.dump({"command": "rocprofv3 --pmc <set> -- python tools/kbench.py $WHAT (averages over all launches of a kernel in the run: every variant row of "
                      "the micro-benchmark launches it)", "kernels": out}, open('$R/gpurun_out/${TAG}_pmc_kbench_${WHAT}_counters.json', 'w'), indent=1, sort_keys=True)
for k, d in sorted(out.items()): print(k[:70], {c: (round(d[c], 3) if isinstance(d.get(c), float) else d.get(c)) for c in ('avg_us', 'calls', 'eff_clock_GHz', 'mfma_busy_frac', 'l2_hit_rate', 'hbm_traffic_MB')})
PY
