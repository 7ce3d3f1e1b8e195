#!/bin/bash
# development: walk-queue kernel time per unit for (LT, LQ)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in ${CFGS:-"1e7 1e7" "1.25e6 1e7" "1e7 1.25e6"}; do
  rm -rf $R/gpurun_out/prof_tmp
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_tmp -o run --output-format csv -- python3 $R/scripts/dev_unit2.py $cfg > $R/gpurun_out/walk_probe.log 2>&1
  grep "^LT" $R/gpurun_out/walk_probe.log | tail -1 | cut -c1-220
  f=$(find $R/gpurun_out/prof_tmp -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k4_extend_hits' in r['Name'] or '1280u' in r['Name']:
        print('   ', r['Name'][:40], 'calls', r['Calls'], 'avg us %.1f' % (float(r['AverageNs']) / 1e3))
PY
done
rm -rf $R/gpurun_out/prof_tmp
