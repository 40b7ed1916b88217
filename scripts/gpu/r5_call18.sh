#!/bin/bash
# round 5, call 18: counters of the long-context attention kernels at depth 16384 (separate --pmc passes, no tracing)
set -u
O=gpurun_out/r5_call18; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P1="FETCH_SIZE GRBM_GUI_ACTIVE"
P2="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --pmc $P --output-format csv -d $R/$O/p$i -o p -- python $R/scripts/depth_only.py llama-3-8b 8 16384 18 > $R/$O/p$i.log 2>&1; echo "pass $i rc=$?" )
  python scripts/pmc_table.py $O/p$i attn_ > $O/pmc_p$i.csv 2>> $O/p$i.log
  find $O/p$i -name "*.csv" -size +2M -delete
done
cat $O/pmc_p*.csv | cut -c1-330
