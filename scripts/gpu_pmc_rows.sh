#!/bin/bash
# development: per-dispatch rows of one PMC counter for the K34 launches of one C4 unit
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_tmp
rocprofv3 --kernel-trace --pmc ${1:-FETCH_SIZE} -d $R/gpurun_out/pmc_tmp -o run --output-format csv -- python3 $R/scripts/dev_unit.py 1e7 1000 > /dev/null 2>&1
f=$(find $R/gpurun_out/pmc_tmp -name "*counter_collection.csv" | head -1)
head -1 "$f" | cut -c1-300
grep "k34_scan" "$f" | cut -c1-260 | head -12
rm -rf $R/gpurun_out/pmc_tmp
