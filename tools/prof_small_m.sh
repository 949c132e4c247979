#!/bin/bash
# step timeline of the headline loop at a small inducing count:  tools/prof_small_m.sh <m> [kind]
R=$GRAFT_REPO_ROOT
m=${1:-32}; kind=${2:-rbf}
O=$R/gpurun_out/small_m; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_sm
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_sm -o sm -- python3 $R/bench.py --steps 60 --warmup 20 --no-cpu --no-extras --m $m --kind $kind > $O/run_${m}_${kind}.log 2>&1
f=$(find /tmp/p_sm -name "*kernel_trace.csv" | head -1)
cd $R && python3 tools/trace_step.py $f > $O/timeline_${m}_${kind}.txt 2>&1
cat $O/timeline_${m}_${kind}.txt
