#!/bin/bash
# kernel statistics of Q4_0 / Q8_0-f32act decode, one-wavefront kernel (GL3_VLQ=0) vs K-split kernel
set -u
O=${1:-gpurun_out/r3q4prof}; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 0 1; do
  for spec in "2 x" "8 f32act"; do
    set -- $spec
    ( cd /tmp && GL3_VLQ=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/t_${v}_$1 -o k -- python $R/scripts/tg_only.py llama-3-8b 8 $1 64 $2 > $R/$O/tg_${v}_$1.log 2>&1; echo "vlq=$v type=$1 rc=$?"; tail -1 $R/$O/tg_${v}_$1.log )
  done
done
find $O -name "*kernel_trace.csv" -delete
python - <<PY
import csv, glob
for d in sorted(glob.glob("$O/t_*")):
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
    print("==", d)
    for r in csv.DictReader(open(f)):
        if float(r["Percentage"]) > 1.0: print("  %-90s calls %6s avg_us %8.2f  %5.1f%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
