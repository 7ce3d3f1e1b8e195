#!/bin/bash
# SQ counters of the seed-scan kernel K34 (first pass) and of its companion kernels on one C4 unit (10 Mbp x 10 Mbp,
# seed 1000): rocprofv3 PMC in separate passes, kernel-trace only (gpurun refuses PMC with other trace domains).
# Averages per launch -> gpurun_out/${TAG:-r03}_pmc_k34_sq.json (copied to profiles/ by hand).  SQ_*_CYCLES and
# SQ_ACTIVE_INST_* / SQ_WAIT_* count quad-cycles summed over the waves (MI355X_MICROARCH.md, PMC table).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r03}
OUT=$R/gpurun_out/${TAG}_pmc_k34_sq.json
PARTS=$R/gpurun_out/pmc_parts
rm -rf $PARTS && mkdir -p $PARTS
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD" \
           "SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR"; do
  rm -rf $R/gpurun_out/pmc_tmp
  MIMEO_K34_DEBUG=${DBG:-0} rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmc_tmp -o run --output-format csv -- python3 $R/scripts/dev_unit.py 1e7 1000 > /dev/null 2>&1
  f=$(find $R/gpurun_out/pmc_tmp -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" $PARTS/pass$i.csv
  i=$((i+1))
done
rm -rf $R/gpurun_out/pmc_tmp
python3 - $PARTS "$OUT" <<'PY'
import csv, sys, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in sorted(glob.glob(sys.argv[1] + '/pass*.csv')):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'k34_scan' in k:
            name = 'k34_scan_extend (split pass)' if 'true>' in k else 'k34_scan_extend (first pass)'
        elif 'k4_walk_batch' in k or 'k4_extend_hits' in k: name = 'k4_walk_batch (walk queue)'
        elif 'k4_extend_generic' in k: name = 'k4_extend_generic'
        elif 'k4_entropy' in k and 'big' not in k: name = 'k4_entropy'
        else: continue
        acc[name][r['Counter_Name']] += float(r['Counter_Value']); cnt[(name, r['Counter_Name'])] += 1
out = {'what': 'rocprofv3 --kernel-trace --pmc, separate passes, one C4 unit (10 Mbp x 10 Mbp, seed 1000, plus strand) through mimeo_ungapped_hsps, three repeats; averages per launch',
       'kernels': {k: {c: round(v / cnt[(k, c)], 1) for c, v in sorted(acc[k].items())} for k in acc},
       'launches': {k: max(cnt[(k, c)] for c in acc[k]) for k in acc}}
k = out['kernels'].get('k34_scan_extend (first pass)', {})
if k.get('SQ_BUSY_CYCLES') and k.get('SQ_ACTIVE_INST_VALU'):
    out['derived_first_pass'] = {
        'valu_active_share_of_wave_cycles': round(k['SQ_ACTIVE_INST_VALU'] / k['SQ_WAVE_CYCLES'], 4) if k.get('SQ_WAVE_CYCLES') else None,
        'quad_cycles_valu_active_per_valu_inst': round(k['SQ_ACTIVE_INST_VALU'] / k['SQ_INSTS_VALU'], 3) if k.get('SQ_INSTS_VALU') else None,
        'lds_bank_conflict_share_of_lds_active': round(k['SQ_LDS_BANK_CONFLICT'] / k['SQ_LDS_IDX_ACTIVE'], 4) if k.get('SQ_LDS_IDX_ACTIVE') else None,
    }
json.dump(out, open(sys.argv[2], 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $PARTS
