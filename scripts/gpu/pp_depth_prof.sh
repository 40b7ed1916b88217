#!/bin/bash
# Kernel statistics of one batched-prefill chunk of 512 tokens behind DEPTH positions (4-layer Llama-3-8B; llama-bench's pp512 @ d).
#   DEPTH=4096 bash scripts/gpu/pp_depth_prof.sh gpurun_out/ppd
set -u
O=${1:-gpurun_out/ppd}; D=${DEPTH:-4096}; R=$GRAFT_REPO_ROOT
mkdir -p $O; export TMPDIR=/tmp
( cd /tmp && PP_DEPTH=$D timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o k -- python $R/scripts/pp_only.py llama-3-8b 4 8 > $R/$O/pp.log 2>&1; echo rc=$? )
cat $O/pp.log | grep pp512
find $O -name "*kernel_trace.csv" -delete
python - <<PY
import csv, glob
f = glob.glob("$O/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:14]: print("%-110s calls %5s avg %10.1f us  %5s %%" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
