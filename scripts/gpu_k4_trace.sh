#!/bin/bash
# kernel durations of one 10 Mbp x 10 Mbp pair for a K4 variant: bash scripts/gpu_k4_trace.sh <variant> [L]
export TMPDIR=/tmp
V=$1; L=${2:-10e6}
D=$PWD/gpurun_out/k4v$V
mkdir -p $D
export MIMEO_K4_VARIANT=$V
rocprofv3 --kernel-trace --output-format csv -d $D -o t -- python3 scripts/dev_unit.py $L 2 > $D/out.log 2>&1
python3 - "$D" <<'PY'
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1] + '/t_kernel_trace.csv')))
agg = collections.OrderedDict()
for r in rows:
    m = re.search(r'mimeo::(\w+)', r['Kernel_Name'])
    k = m.group(1) if m else 'other'
    a = agg.setdefault(k, [0, 0.0, r['VGPR_Count'], r['LDS_Block_Size']])
    a[0] += 1; a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for k, (n, t, vg, lds) in sorted(agg.items(), key=lambda kv: (kv[0] != "k4_extend_hits", -kv[1][1]))[:8]:
    print('%-24s calls %4d avg %9.1f us  vgpr %s lds %s' % (k, n, t / n, vg, lds))
PY
