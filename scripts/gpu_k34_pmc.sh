#!/bin/bash
# development: SQ counters of the fused kernel on one C4 unit (separate passes, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD"; do
  rm -rf $R/gpurun_out/pmc_tmp
  MIMEO_K34_CFG=${CFG:-2} MIMEO_K34_DEBUG=${DBG:-0} rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmc_tmp -o run --output-format csv -- python3 $R/scripts/dev_unit.py 1e7 1000 > /dev/null 2>&1
  f=$(find $R/gpurun_out/pmc_tmp -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'][:60]
    if not any(x in k for x in ('k34_scan', 'k4_extend_hits', 'k4_entropy', 'k4_extend_generic')): continue
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
for k in acc:
    print(k, {c: '%.4g' % (v / cnt[(k, c)]) for c, v in acc[k].items()})
PY
done
rm -rf $R/gpurun_out/pmc_tmp
