#!/bin/bash
# usage: tools/prof.sh <name> <python script + args...>   -- rocprofv3 kernel-trace + stats of one program, CSV output,
# summaries copied to gpurun_out/<name>_{kernel_stats.csv,kernel_trace.csv}.  (The program goes directly after `--`.)
set -e
name=$1; shift
out=/tmp/prof_$name
rm -rf $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $name -- python3 "$@" > $GRAFT_REPO_ROOT/gpurun_out/${name}.log 2>&1 || true
cd $GRAFT_REPO_ROOT
f=$(find $out -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/${name}_kernel_stats.csv
f=$(find $out -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/${name}_kernel_trace.csv
exit 0
