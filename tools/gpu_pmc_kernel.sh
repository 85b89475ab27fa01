#!/bin/bash
# SQ counters of ONE kernel family driven by tools/kbench.py (separate --pmc passes, no tracing).
#   gpurun --timeout 600 -- 'bash tools/gpu_pmc_kernel.sh k1 k_consensus_params_circ TAG'
R=${GRAFT_REPO_ROOT:-/root/repo}
WHAT=${1:-k1}; KERN=${2:-k_consensus_params_circ}; TAG=${3:-r02}
mkdir -p $R/gpurun_out/pmck
cd /tmp && export TMPDIR=/tmp
i=0
for c in ${RCMARL_PMC_SETS:-"SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_IFETCH SQ_ACTIVE_INST_SCA" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_MISC"}; do
  i=$((i+1))
  RCMARL_KBENCH_ONLY=${RCMARL_KBENCH_ONLY:-18} timeout 200 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmck -o ${TAG}_p$i -- python $R/tools/kbench.py $WHAT > $R/gpurun_out/pmck/${TAG}_p$i.log 2>&1
  tail -2 $R/gpurun_out/pmck/${TAG}_p$i.log | cut -c1-160
done
python - <<PY
import csv, glob, collections, json
tot = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob('$R/gpurun_out/pmck/${TAG}_p*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if '$KERN' not in r['Kernel_Name']: continue
        tot[r['Counter_Name']] += float(r['Counter_Value']); cnt[r['Counter_Name']] += 1
out = {k: tot[k] / cnt[k] for k in tot}
out['launches_seen'] = dict(cnt)
json.dump(out, open('$R/gpurun_out/${TAG}_sq_${KERN}.json', 'w'), indent=1)
for k in sorted(out): print(k, out[k])
PY
