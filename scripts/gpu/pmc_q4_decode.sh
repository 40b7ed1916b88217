#!/bin/bash
# SQ counters of the Q4_0 decode matvecs, one-wavefront kernel (GL3_VLQ=0) vs K-split kernel
set -u
O=${1:-gpurun_out/r3q4pmc}; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 0 1; do
  ( cd /tmp && GL3_VLQ=$v timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES SQ_ACTIVE_INST_LDS --output-format csv -d $R/$O/p_$v -o p -- python $R/scripts/tg_only.py llama-3-8b 4 2 16 > $R/$O/tg_$v.log 2>&1; echo "vlq=$v rc=$?"; tail -1 $R/$O/tg_$v.log )
done
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$O/p_*")):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "matvec_vl" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][:60], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", d)
    for k, v in sorted(acc.items()):
        m = {c: sum(x) / len(x) for c, x in v.items()}
        wc = m.get("SQ_WAVE_CYCLES", 1)
        print(" ", k, "n", len(next(iter(v.values()))), "busy_cyc/32", round(m.get("SQ_BUSY_CYCLES", 0) / 32), "waves", round(m.get("SQ_WAVES", 0)), "valu_insts", round(m.get("SQ_INSTS_VALU", 0)),
              "valu_active%%", round(100 * m.get("SQ_ACTIVE_INST_VALU", 0) / wc, 1), "issue_stall%%", round(100 * m.get("SQ_WAIT_INST_ANY", 0) / wc, 1), "parked%%", round(100 * m.get("SQ_WAIT_ANY", 0) / wc, 1),
              "lds_active%%", round(100 * m.get("SQ_ACTIVE_INST_LDS", 0) / wc, 1), "wave_cycles", round(wc))
PY
