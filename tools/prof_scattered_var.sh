#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/scvar
timeout -k 10 200 python3 $R/tools/diag_scattered_var.py > $R/gpurun_out/scvar/plain.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/scvar/prof -- python3 $R/tools/diag_scattered_var.py > $R/gpurun_out/scvar/prof.log 2>&1
f=$(find $R/gpurun_out/scvar/prof -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' > $R/gpurun_out/scvar/summary.txt
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# per kernel name: list of durations of the big ones
big = collections.defaultdict(list)
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if d > 300: big[r["Kernel_Name"].split("(")[0]].append(round(d))
for k, v in big.items(): print(k, len(v), v[:60])
PY
find $R/gpurun_out/scvar/prof -name "*.csv" -delete
cat $R/gpurun_out/scvar/plain.log $R/gpurun_out/scvar/prof.log | grep -v amdgpu.ids; cat $R/gpurun_out/scvar/summary.txt | cut -c1-700
