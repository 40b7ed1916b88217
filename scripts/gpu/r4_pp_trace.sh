#!/bin/bash
# r4: rocprofv3 kernel statistics of pp512 on N layers of a config ($2 model, $3 layers); extra env in $4
set -u
O=$1; M=${2:-llama-3-8b}; NL=${3:-4}; V=${4:-}; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( cd /tmp && env $V timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o k -- python $R/scripts/pp_only.py $M $NL 8 > $R/$O/pp.log 2>&1; echo rc=$? )
tail -1 $O/pp.log
python - <<PY
import csv, glob
f = glob.glob("$O/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
with open("$O/kernel_stats.csv", "w") as fo:
    fo.write("Name,Calls,TotalDurationNs,AverageNs\n")
    for r in rows[:25]:
        fo.write('"%s",%s,%s,%s\n' % (r["Name"][:100], r["Calls"], r["TotalDurationNs"], r["AverageNs"]))
        print("%-80s calls %5s avg %9.1f us total %8.2f ms" % (r["Name"][:80], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
find $O/trace -name "*kernel_trace.csv" -delete
