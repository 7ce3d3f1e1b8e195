#!/bin/bash
# time the K4a kernel variants on one 5 Mbp x 5 Mbp unit (rocprof kernel trace)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 1 2 3; do
  MIMEO_K4_VARIANT=$v timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/k4v_$v -- python scripts/dev_unit.py 5e6 1 > gpurun_out/k4v_$v.log 2>&1
  f=$(find gpurun_out/k4v_$v -name "*kernel_stats.csv" | head -1)
  echo "variant $v: $(grep -E 'k4_extend_(hits|generic)' $f | awk -F, '{print $(NF-6), $(NF-5), $(NF-4)}' | tr '\n' ' ')"
done
