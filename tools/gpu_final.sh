#!/bin/bash
# The evidence run of a round on ONE box: counters of the lattice GEMMs and of the wide GEMM, the effective clock next to the MFMA
# micro-benchmark, the GPU suite, the bench line, rocprofv3 kernel stats and PMC traffic of the headline workload.
#   gpurun --timeout 2400 -- 'bash tools/gpu_final.sh TAG [notests]'
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
TAG=${1:-r05}
[ -x tools/micro/lat_bench ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/micro/lat_bench.hip -o tools/micro/lat_bench -ldl 2> /dev/null
[ -x tools/micro/mfma_peak ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/micro/mfma_peak.hip -o tools/micro/mfma_peak 2> /dev/null
echo "== lattice GEMM counters (both shapes, final build)"
bash tools/gpu_pmc_latbench.sh ${TAG} both X=1 > gpurun_out/${TAG}_pmc_latbench.txt 2>&1
grep -E "^k_lat|eff_clock|mfma_busy|l2_hit|avg_us" gpurun_out/${TAG}_pmc_latbench.txt
echo "== lattice GEMM times (both shapes)"
tools/micro/lat_bench resilient-consensus-based-marl_amd/lib/librcmarl_hip.so both X=0 X=1 2>&1 | tee gpurun_out/${TAG}_lat_bench_final.txt | grep -E "^(fwd|bwd)"
echo "== MFMA micro-benchmark with its clock"
if [ -x tools/micro/mfma_peak ]; then
  ( cd /tmp && export TMPDIR=/tmp
    timeout 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmcmf -o ${TAG}_mf -- $R/tools/micro/mfma_peak > $R/gpurun_out/${TAG}_mfma_peak_pmc.log 2>&1
    timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pmcmf -o ${TAG}_mft -- $R/tools/micro/mfma_peak > $R/gpurun_out/${TAG}_mfma_peak_trace.log 2>&1 )
  tools/micro/mfma_peak 2>&1 | tail -6 | tee gpurun_out/${TAG}_mfma_peak.txt
  python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter); dur = {}
for f in glob.glob('$R/gpurun_out/pmcmf/${TAG}_mf*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        tot[r['Kernel_Name']][r['Counter_Name']] += float(r['Counter_Value']); cnt[r['Kernel_Name']][r['Counter_Name']] += 1
for f in glob.glob('$R/gpurun_out/pmcmf/${TAG}_mft*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        dur[r['Name']] = float(r['AverageNs'])
with open('$R/gpurun_out/${TAG}_mfma_peak_clock.txt', 'w') as out:
    for k in tot:
        if k not in dur: continue
        gui = tot[k]['GRBM_GUI_ACTIVE'] / max(cnt[k]['GRBM_GUI_ACTIVE'], 1)
        busy = tot[k]['SQ_VALU_MFMA_BUSY_CYCLES'] / max(cnt[k]['SQ_VALU_MFMA_BUSY_CYCLES'], 1)
        clk = gui / 8 / dur[k]
        line = "%s: avg %.1f us, effective clock %.3f GHz (GRBM_GUI_ACTIVE / 8 / duration), matrix pipe busy %.3f of the cycles" % (k[:60], dur[k] / 1e3, clk, busy / (1024 * clk * dur[k]))
        print(line); out.write(line + "\n")
PY
fi
echo "== k_wgemm16 counters (tools/kbench.py wide)"
mkdir -p gpurun_out/pmcw
i=0
for c in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_WRITE_sum"; do
  i=$((i+1))
  ( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmcw -o ${TAG}_w$i -- python $R/tools/kbench.py wide > $R/gpurun_out/pmcw/${TAG}_w$i.log 2>&1 )
done
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pmcw -o ${TAG}_wt -- python $R/tools/kbench.py wide > $R/gpurun_out/pmcw/${TAG}_wt.log 2>&1 )
python3 - <<PY
import csv, glob, collections, json, re
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter); dur = {}
key = lambda n: (re.search(r'(k_wgemm16<[^>]*>|k_wgemm<[^>]*>)', n) or [None])[0]
for f in glob.glob('$R/gpurun_out/pmcw/${TAG}_w[0-9]*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = key(r['Kernel_Name'])
        if k: tot[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k][r['Counter_Name']] += 1
for f in glob.glob('$R/gpurun_out/pmcw/${TAG}_wt*kernel_stats.csv'):
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
    out[k] = d
json.dump(out, open('$R/gpurun_out/${TAG}_pmc_kbench_wide_counters.json', 'w'), indent=1, sort_keys=True)
for k, d in out.items(): print(k, {c: d.get(c) for c in ('avg_us', 'eff_clock_GHz', 'mfma_busy_frac', 'l2_hit_rate')})
PY
if [ "$2" != "notests" ]; then W="tests"; else W=""; fi
bash tools/gpu_visit.sh ${TAG} $W bench prof:cfg4_shard pmc:cfg4_shard pmc:target_N256_H1 prof:cfg2_single
