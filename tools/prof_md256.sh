#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/md256
timeout -k 10 200 python3 $R/tools/diag_md256.py rbf 256 > $R/gpurun_out/md256/rbf.log 2>&1
timeout -k 10 200 python3 $R/tools/diag_md256.py matern32 256 > $R/gpurun_out/md256/m32.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/md256/prof -- python3 $R/tools/diag_md256.py rbf 256 > $R/gpurun_out/md256/run.log 2>&1
f=$(find $R/gpurun_out/md256/prof -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/md256_kernel_stats.csv
find $R/gpurun_out/md256/prof -name "*kernel_trace.csv" -delete
cat $R/gpurun_out/md256/rbf.log $R/gpurun_out/md256/m32.log
head -12 $R/gpurun_out/md256_kernel_stats.csv | cut -c1-150
