#!/bin/bash
# kernel stats of the masked step at m_d = 128 (M = 16384)
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/mbprof
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/mbprof -- python3 $R/tools/time_masked_big.py 128 > $R/gpurun_out/mbprof/run.log 2>&1
f=$(find $R/gpurun_out/mbprof -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/mbprof_kernel_stats.csv
find $R/gpurun_out/mbprof -name "*kernel_trace.csv" -delete
head -25 $R/gpurun_out/mbprof_kernel_stats.csv
