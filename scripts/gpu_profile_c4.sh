#!/bin/bash
# Profiles behind the default bench line (run on the GPU box from the repo root):
#   1. rocprofv3 --kernel-trace --stats of the bench command (short) -> gpurun_out/prof_c4/
#   2. two separate PMC passes (FETCH_SIZE, WRITE_SIZE) over one 10 Mbp x 10 Mbp pair -> gpurun_out/pmc_*/
set -e
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/prof_c4 gpurun_out/pmc_fetch gpurun_out/pmc_write
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_c4 -o c4 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_c4/bench_line.json 2> gpurun_out/prof_c4/err.log
python3 scripts/rocpd_stats.py $(ls gpurun_out/prof_c4/*results.db | head -1) > gpurun_out/prof_c4/kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -o f -- python3 scripts/dev_unit.py 10e6 1 > gpurun_out/pmc_fetch/out.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -o w -- python3 scripts/dev_unit.py 10e6 1 > gpurun_out/pmc_write/out.log 2>&1
python3 - <<'PY'
import csv, glob, json, collections
out = {}
for name, d in (('FETCH_SIZE', 'gpurun_out/pmc_fetch'), ('WRITE_SIZE', 'gpurun_out/pmc_write')):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            k = row['Kernel_Name']
            if 'k3_join' not in k:
                continue
            short = 'k3_join_fill' if 'k3_join_fill' in k else 'k3_join_count' if 'k3_join_count' in k else k[:40]
            if row['Counter_Name'] == name:
                out.setdefault(short, collections.OrderedDict()).setdefault(name + '_KB', []).append(round(float(row['Counter_Value'])))
json.dump(out, open('gpurun_out/pmc_c4_unit.json', 'w'), indent=1)
print(json.dumps(out))
PY
