#!/bin/bash
# rocprofv3 kernel trace of the default bench (C4 rows); summary -> gpurun_out/<tag>_kernel_stats.csv
tag=${1:-r03_bench_c4_rows}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_tmp -o run --output-format csv -- python3 $R/bench.py --steps ${STEPS:-4} --warmup 1 --no-cpu-baseline ${EXTRA} > $R/gpurun_out/${tag}_line.json 2> $R/gpurun_out/${tag}_stderr.log
f=$(find $R/gpurun_out/prof_tmp -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/${tag}_kernel_stats.csv
head -25 $R/gpurun_out/${tag}_kernel_stats.csv | cut -c1-200
rm -rf $R/gpurun_out/prof_tmp
