#!/bin/bash
# kernel statistics of Q4_0 / Q8_0-f32act / F16 decode (8 layers at the 8B / 1B shapes)
set -u
O=${1:-gpurun_out/r3q4prof}; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for spec in "llama-3-8b 2 x" "llama-3-8b 8 f32act" "llama-3.2-1b 1 x"; do
  set -- $spec
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/t_$1_$2 -o k -- python $R/scripts/tg_only.py $1 8 $2 64 $3 > $R/$O/tg_$1_$2.log 2>&1; echo "$1 type=$2 rc=$?"; tail -1 $R/$O/tg_$1_$2.log )
done
find $O -name "*kernel_trace.csv" -delete
python - <<PY
import csv, glob
for d in sorted(glob.glob("$O/t_*")):
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
    print("==", d)
    for r in csv.DictReader(open(f)):
        if "gl3" in r["Name"] and float(r["Percentage"]) > 0.3: print("  %-100s calls %6s avg_us %8.2f  %5.1f%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
