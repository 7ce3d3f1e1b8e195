#!/bin/bash
# development: kernel times of one C4 unit with different grid sizes of the walk-queue kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for b in 128 256 512 1024; do
  rm -rf $R/gpurun_out/qw_tmp
  MIMEO_QW_BLOCKS=$b rocprofv3 --kernel-trace --stats -d $R/gpurun_out/qw_tmp -o run --output-format csv -- python3 $R/scripts/dev_unit.py 1e7 1000 > /dev/null 2>&1
  f=$(find $R/gpurun_out/qw_tmp -name "*kernel_stats.csv" | head -1)
  echo "blocks $b: $(grep -E 'k4_extend_hits|k34_scan' $f | sed 's/(.*)//' | cut -d, -f1,4 | tr '\n' ' ')"
done
rm -rf $R/gpurun_out/qw_tmp
