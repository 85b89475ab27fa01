#!/bin/bash
# Hardware counters of the lattice GEMM kernels driven by tools/micro/lat_bench (no Python in the profiled process: a pass takes
# seconds).  Separate --pmc passes (no tracing in them), then one kernel-trace pass for the durations.
#   gpurun --timeout 600 -- 'bash tools/gpu_pmc_latbench.sh TAG fwd|bwd|both CONFIG...'   -> gpurun_out/TAG_pmc_latbench.json
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
L=$R/resilient-consensus-based-marl_amd/lib/librcmarl_hip.so
mkdir -p $R/gpurun_out/pmclb
[ -x $R/tools/micro/lat_bench ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 $R/tools/micro/lat_bench.hip -o $R/tools/micro/lat_bench -ldl 2> /dev/null
cd /tmp && export TMPDIR=/tmp
export LB_ITERS=${LB_ITERS:-5}
i=0
for c in ${RCMARL_PMC_SETS:-"GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_WRITE_sum" "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"}; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmclb -o ${TAG}_p$i -- $R/tools/micro/lat_bench $L "$@" > $R/gpurun_out/pmclb/${TAG}_p$i.log 2>&1
  tail -1 $R/gpurun_out/pmclb/${TAG}_p$i.log | cut -c1-160
done
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pmclb -o ${TAG}_trace -- $R/tools/micro/lat_bench $L "$@" > $R/gpurun_out/pmclb/${TAG}_trace.log 2>&1
python3 - <<PY
import csv, glob, collections, json, re
def key(n):
    m = re.search(r'(k_lat_\w+)(<[^>]*>)?', n)
    return (m.group(1) + (m.group(2) or '')) if m else None
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for f in glob.glob('$R/gpurun_out/pmclb/${TAG}_p*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = key(r['Kernel_Name'])
        if not k: continue
        tot[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k][r['Counter_Name']] += 1
out = {}
for k in tot:
    out[k] = {c: tot[k][c] / cnt[k][c] for c in tot[k]}
    out[k]['launches_seen'] = max(cnt[k].values())
for f in glob.glob('$R/gpurun_out/pmclb/${TAG}_trace*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        k = key(r['Name'])
        if k and k in out:
            out[k]['avg_us'] = float(r['AverageNs']) / 1e3; out[k]['calls'] = int(r['Calls'])
for k, d in out.items():
    us = d.get('avg_us')
    if us and 'GRBM_GUI_ACTIVE' in d:
        d['eff_clock_GHz'] = d['GRBM_GUI_ACTIVE'] / 8 / (us * 1e3)           # summed over the 8 XCDs
    if us and 'SQ_VALU_MFMA_BUSY_CYCLES' in d and 'eff_clock_GHz' in d:
        d['mfma_busy_frac'] = d['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * d['eff_clock_GHz'] * 1e3 * us)   # busy cycles per SIMD / cycles
    if 'TCC_HIT_sum' in d:
        d['l2_hit_rate'] = d['TCC_HIT_sum'] / max(d['TCC_HIT_sum'] + d['TCC_MISS_sum'], 1)
json.dump(out, open('$R/gpurun_out/${TAG}_pmc_latbench.json', 'w'), indent=1, sort_keys=True)
for k, d in out.items():
    print(k)
    for c in sorted(d): print('   %-32s %s' % (c, d[c]))
PY
