#!/bin/bash
# round 5, call 21: final form of the Q4_0 product GEMM: kernel statistics and SQ counters (2 layers of the 8B shape, pp512)
set -u
O=gpurun_out/r5_call21; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o k -- python $R/scripts/pp_only.py llama-3-8b 2 2 > $R/$O/trace.log 2>&1 )
find $O -name "*kernel_trace.csv" -delete
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"
( cd /tmp && timeout 300 rocprofv3 --pmc $P1 --output-format csv -d $R/$O/p1 -o p -- python $R/scripts/pp_only.py llama-3-8b 2 2 > $R/$O/p1.log 2>&1 )
python scripts/pmc_table.py $O/p1 gemm_vlq_mfma > $O/pmc_p1.csv; find $O/p1 -name "*.csv" -size +2M -delete
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); head -6 "$f" | cut -c1-150; cat $O/pmc_p1.csv | cut -c1-300
