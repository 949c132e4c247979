#!/bin/bash
# One rocprofv3 kernel trace of the headline bench (graph replay) + its step timeline:  tools/prof_step.sh <outdir> [bench args...]
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_step
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_step -o step -- python3 $R/bench.py --steps 60 --warmup 20 --no-cpu --no-extras "$@" > $O/step.log 2>&1
for suf in kernel_stats kernel_trace; do
    f=$(find /tmp/p_step -name "*${suf}.csv" | head -1); if [ -n "$f" ]; then cp $f $O/step_${suf}.csv; fi
done
cd $R
python3 tools/trace_step.py $O/step_kernel_trace.csv > $O/step_timeline.txt 2>&1
cat $O/step_timeline.txt
