#!/bin/bash
# Collects every profile the docs / bench.py cite, on the GPU box:  tools/profile_round.sh <tag>   (e.g. r2)
# Outputs under gpurun_out/<tag>_prof/ ; copy what is to be judged into profiles/.
# rocprofv3: the program goes directly after `--`; PMC passes run alone (kernel-trace/stats only).
set -u
tag=${1:-r2}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${tag}_prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {   # run <name> <rocprof extra args...> -- <program args...>
    local name=$1; shift
    local extra=()
    while [ "$1" != "--" ]; do extra+=("$1"); shift; done; shift
    rm -rf /tmp/p_$name
    timeout -k 10 500 rocprofv3 "${extra[@]}" --kernel-trace --stats --output-format csv -d /tmp/p_$name -o $name -- python3 "$@" > $O/$name.log 2>&1
    for suf in kernel_stats kernel_trace counter_collection; do
        f=$(find /tmp/p_$name -name "*${suf}.csv" | head -1); if [ -n "$f" ]; then cp $f $O/${name}_${suf}.csv; fi
    done
    return 0
}
B="$R/bench.py --steps 60 --warmup 20 --no-cpu --no-extras"
run step -- $B                                                     &&
run step_m32 -- $B --kind matern32                                  &&
VGGP_NO_GRAPH=1 run pmc_fetch --pmc FETCH_SIZE -- $B                &&
VGGP_NO_GRAPH=1 run pmc_write --pmc WRITE_SIZE -- $B                &&
run pmc_calib --pmc FETCH_SIZE -- $R/tools/pmc_calib.py             &&
VGGP_NO_GRAPH=1 run pmc_fetch_slab --pmc FETCH_SIZE -- $B --n1 4096 --n2-local 1024 &&
VGGP_NO_GRAPH=1 run pmc_write_slab --pmc WRITE_SIZE -- $B --n1 4096 --n2-local 1024 &&
run factor -- $R/tools/time_factor.py                               &&
run factor_fetch --pmc FETCH_SIZE -- $R/tools/time_factor.py        &&
run factor_write --pmc WRITE_SIZE -- $R/tools/time_factor.py        &&
run trsm_kron -- $R/tools/time_trsm.py                              &&
run masked -- $R/bench.py --masked --n 2048 --m 32 --steps 20 --warmup 5 --no-cpu &&
run md256 -- $R/tools/time_md256.py rbf matern32                    &&
run masked_iter -- $R/tools/time_masked_iter.py                     &&
run families -- $R/tools/time_families.py b1 vff "matern52 points m=128"
cd $R
python3 tools/trace_step.py $O/step_kernel_trace.csv > $O/step_timeline.txt 2>&1
python3 tools/trace_step.py $O/step_m32_kernel_trace.csv > $O/step_m32_timeline.txt 2>&1
CALIB_BYTES=134217728 python3 tools/pmc_traffic.py $O/pmc_fetch_counter_collection.csv $O/pmc_write_counter_collection.csv $O/pmc_calib_counter_collection.csv $O/pmc_traffic.json 1024 1024 128 > $O/pmc_traffic.log 2>&1
CALIB_BYTES=134217728 python3 tools/pmc_traffic.py $O/pmc_fetch_slab_counter_collection.csv $O/pmc_write_slab_counter_collection.csv $O/pmc_calib_counter_collection.csv $O/pmc_traffic_slab_1024x4096.json 4096 1024 128 > $O/pmc_traffic_slab.log 2>&1
ls $O | head -50
