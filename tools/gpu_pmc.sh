#!/bin/bash
# PMC passes (separate from any tracing, as gpurun requires).  gpurun --timeout 900 -- 'bash tools/gpu_pmc.sh <what> <counters...>'
R=${GRAFT_REPO_ROOT:-/root/repo}
what=$1; shift
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc -o ${what}_$i -- python $R/tools/kbench.py $what > $R/gpurun_out/pmc/${what}_$i.log 2>&1
  tail -2 $R/gpurun_out/pmc/${what}_$i.log
done
ls $R/gpurun_out/pmc | head -30
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob('$R/gpurun_out/pmc/${what}_*counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:60]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); 
    print(f.split('/')[-1])
    for k, v in agg.items():
        if 'rcmarl' in k or 'k_' in k or 'fast' in k:
            print('  ', k, {a: '%.4g' % b for a, b in v.items()})
PY
