#!/bin/bash
# SQ stall / LDS breakdown of the batched-prefill kernels (pp512, 8 layers of the 8B shape): separate --pmc passes, CSV, bounded
set -u
O=${1:-gpurun_out/pp_pmc}; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $R/$O/pmc1 -o p -- python $R/scripts/pp_only.py llama-3-8b 4 > $R/$O/p1.log 2>&1; echo rc=$? )
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $R/$O/pmc2 -o p -- python $R/scripts/pp_only.py llama-3-8b 4 > $R/$O/p2.log 2>&1; echo rc=$? )
python3 - $O <<'PY'
import csv, collections, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/pmc*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "pf_" in r["Kernel_Name"]: agg[r["Kernel_Name"][:70] + " grid=" + r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(k); print("   " + "  ".join("%s=%.3g" % (c, sum(x) / len(x)) for c, x in v.items()))
PY
