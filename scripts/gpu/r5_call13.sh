#!/bin/bash
# round 5, call 13: Q4_0 batched GEMM on the matrix cores: kernel trace + SQ counters (2 layers, pp512)
set -u
O=gpurun_out/r5_call13; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o k -- python $R/scripts/pp_only.py llama-3-8b 2 2 > $R/$O/trace.log 2>&1 )
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-180
python - <<'PY'
import csv, glob, collections
fs = glob.glob("gpurun_out/r5_call13/trace/**/*kernel_trace.csv", recursive=True)
if fs:
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "gemm_vlq_mfma" in r["Kernel_Name"]:
            d[(r["Kernel_Name"][:70], r.get("Grid_Size_X", r.get("Grid_Size")))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in sorted(d.items()): print(k, len(v), "avg us %.1f" % (sum(v) / len(v)))
PY
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_MISC"
P3="SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --pmc $P --output-format csv -d $R/$O/p$i -o p -- python $R/scripts/pp_only.py llama-3-8b 2 2 > $R/$O/p$i.log 2>&1; echo "pass $i rc=$?" )
  python scripts/pmc_table.py $O/p$i gemm_vlq_mfma > $O/pmc_p$i.csv 2>> $O/p$i.log
  find $O/p$i -name "*.csv" -size +2M -delete
done
cat $O/pmc_p*.csv
