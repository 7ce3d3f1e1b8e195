#!/bin/bash
# HBM traffic of the seed-scan kernel (K34) and of its companion kernels on one C4 unit and one C2 unit: rocprofv3 PMC,
# FETCH_SIZE and WRITE_SIZE in separate passes (they do not fit one pass), kernel-trace only.  FETCH_SIZE counts
# 64-byte requests where wide coalesced reads issue 128-byte ones on gfx950 (MI355X_MICROARCH.md, HBM): doubled in
# the summary, WRITE_SIZE taken as is.  The split pass of K34 (a launch that finds no listed tile on these units) is left out of
# the average.  -> gpurun_out/${TAG:-r03b}_pmc_seed_scan.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${TAG:-r03b}_pmc_seed_scan.json
echo "{" > $OUT
first=1
for unit in "c4_unit 1e7 1000" "c2_unit 5e6 50"; do
  set -- $unit
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/pmc_tmp
    rocprofv3 --kernel-trace --pmc $ctr -d $R/gpurun_out/pmc_tmp -o run --output-format csv -- python3 $R/scripts/dev_unit.py $2 $3 > /dev/null 2>&1
    f=$(find $R/gpurun_out/pmc_tmp -name "*counter_collection.csv" | head -1)
    [ $first -eq 1 ] || echo "," >> $OUT
    first=0
    python3 - "$f" "$1" "$ctr" >> $OUT <<'PY'
import csv, sys, collections, json
acc = collections.defaultdict(float); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name']
    name = 'k34_scan_extend' if ('k34_scan' in k and 'true>' not in k) else ('k4_walk_batch (walk queue)' if ('k4_walk_batch' in k or 'k4_extend_hits' in k) else None)
    if name is None or r['Counter_Name'] != sys.argv[3]: continue
    acc[name] += float(r['Counter_Value']); cnt[name] += 1
print('"%s_%s_KB_per_launch": %s' % (sys.argv[2], sys.argv[3], json.dumps({k: round(v / cnt[k], 1) for k, v in acc.items()})), end='')
PY
  done
done
echo "}" >> $OUT
rm -rf $R/gpurun_out/pmc_tmp
cat $OUT
