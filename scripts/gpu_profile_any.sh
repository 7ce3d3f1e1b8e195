#!/bin/bash
# rocprofv3 kernel trace of a bench workload; summary -> gpurun_out/<tag>_kernel_stats.csv      usage: gpu_profile_any.sh <tag> <bench args...>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_tmp -o run --output-format csv -- python3 $R/bench.py "$@" --no-cpu-baseline > $R/gpurun_out/${tag}_line.json 2> $R/gpurun_out/${tag}_stderr.log
f=$(find $R/gpurun_out/prof_tmp -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/${tag}_kernel_stats.csv
head -30 $R/gpurun_out/${tag}_kernel_stats.csv | cut -c1-220
rm -rf $R/gpurun_out/prof_tmp
