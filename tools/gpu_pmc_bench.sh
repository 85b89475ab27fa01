#!/bin/bash
# HBM traffic of bench.py's kernels from rocprofv3 PMC passes (separate passes, no tracing, as gpurun requires).
#   gpurun --timeout 900 -- 'bash tools/gpu_pmc_bench.sh [workload] [tag]'
# Writes gpurun_out/<tag>_pmc_<workload>.json: per-launch FETCH_SIZE (doubled, MI355X_MICROARCH.md HBM section) + WRITE_SIZE.
R=${GRAFT_REPO_ROOT:-/root/repo}
W=${1:-cfg4_shard}
TAG=${2:-r01}
mkdir -p $R/gpurun_out/pmcb
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmcb -o ${W}_$c -- python $R/bench.py --steps 1 --warmup 1 --workload $W --no-cpu-baseline --no-kernel-timing --no-extra > $R/gpurun_out/pmcb/${W}_$c.log 2>&1
  tail -1 $R/gpurun_out/pmcb/${W}_$c.log | cut -c1-200
done
python - <<PY
import csv, glob, collections, json
names = {"k_lat_forward<true, true, true>": "layer1_forward_lattice_pk", "k_lat_fit": "fit_fused_lattice", "k_lat_forward": "layer1_forward_lattice", "k_lat_backward_sgd": "layer1_backward_sgd_lattice", "k_mid_fit_v3<20, true": "mid_fit_lattice", "k_mid_fit_v8<20, true": "mid_fit_lattice", "k_mid_fit_v5<20, true": "mid_fit_lattice_fixup", "k_mid_fit_v5<20, false": "mid_fit",
         "k_mid_fit_mfma<20, true>": "mid_fit_lattice", "k_w1_split": "w1_split", "k_consensus_params_circ": "consensus_params_circulant", "k_consensus_params": "consensus_params",
         "k_consensus_head<": "consensus_head", "k_consensus_head_mx<": "consensus_head", "k_mid_value_mx<": "mid_value",
         "k_lat_forward<true, true, true>": "layer1_forward_lattice_pk", "k_pk_forward2": "pk_forward2", "k_pk_backward_data": "pk_backward_data",
         "k_pk_backward_w2": "pk_backward_w2", "k_pk_pack_w2": "pk_pack_w2", "k_mid_value": "mid_value", "k_lattice_encode": "lattice_encode",
         "fast::k_fwd": "layer1_forward", "fast::k_bwd<1, (anonymous namespace)::fast::ApplySgd>": "layer1_backward_sgd",
         "k_mid_fit_v3<20, false": "mid_fit", "k_rollout_step_ep": "rollout_step_episodes"}
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.Counter())
for f in glob.glob('$R/gpurun_out/pmcb/${W}_*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = next((v for p, v in names.items() if p in r['Kernel_Name']), None)
        if k is None: continue
        tot[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k][r['Counter_Name']] += 1
out = {"workload": "$W", "command": "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} -- python bench.py --steps 1 --warmup 1 --workload $W --no-cpu-baseline --no-kernel-timing",
       "units": "FETCH_SIZE/WRITE_SIZE are KiB per dispatch; FETCH doubled (gfx950 counts 64 B per 128-B request)",
       "per_launch": {}, "traffic_bytes_per_launch": {}}
for k in tot:
    f = tot[k].get('FETCH_SIZE', 0.0) / max(cnt[k].get('FETCH_SIZE', 1), 1); w = tot[k].get('WRITE_SIZE', 0.0) / max(cnt[k].get('WRITE_SIZE', 1), 1)
    out["per_launch"][k] = {"FETCH_SIZE_KiB_raw": f, "WRITE_SIZE_KiB": w, "launches_seen": int(cnt[k].get('FETCH_SIZE', 0))}
    out["traffic_bytes_per_launch"][k] = (2.0 * f + w) * 1024.0
json.dump(out, open('$R/gpurun_out/${TAG}_pmc_$W.json', 'w'), indent=1)
for k, v in sorted(out["traffic_bytes_per_launch"].items(), key=lambda kv: -kv[1]): print("%-32s %8.1f MB/launch  %s" % (k, v / 1e6, out["per_launch"][k]))
PY
