#!/bin/bash
# rocprofv3 kernel statistics of the static-batched decode step (Qwen3-4B, B = 32): CSV output, bounded by timeout.
set -u
O=${1:-gpurun_out/prof_bd}; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o bd -- python $R/scripts/bd_only.py ${2:-qwen3-4b} ${3:-32} 16 > $R/$O/bd.log 2> $R/$O/bd.err; echo rc=$? )
cat $O/bd.log
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); grep -v "at::native\|rocclr" $f | head -16 | cut -c1-200
