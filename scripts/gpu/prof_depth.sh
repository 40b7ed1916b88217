#!/bin/bash
# kernel statistics of decode at depth (8 layers of the 8B shape): rocprofv3 --kernel-trace --stats, csv
set -u
O=gpurun_out/${OUT:-prof_depth}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for d in ${DEPTHS:-4096 16384}; do
  ( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/d$d -- python $R/scripts/depth_only.py llama-3-8b 8 $d 18 ) > $R/$O/run_d$d.log 2>&1
  tail -2 $R/$O/run_d$d.log
  f=$(find $R/$O/d$d -name "*kernel_stats.csv" | head -1)
  echo "== depth $d: $f"; python - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
for r in rows[:40]:
    print("%-90s calls %6s avg %10.1f ns  %5.1f %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]), float(r["Percentage"])))
PY
  cp $f $R/$O/kernel_stats_d$d.csv
  rm -rf $R/$O/d$d
done
