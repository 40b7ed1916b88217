#!/bin/bash
# rocprofv3 kernel statistics of the decode path (tg only): CSV output, bounded by timeout (rocprofv3 with the default
# rocpd output hung after finalisation once and cost 15 GPU-minutes).
set -u
O=${1:-gpurun_out/prof_tg}; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o tg -- python $R/bench.py --steps 2 --no-pp --no-cpu-baseline > $R/$O/bench.json 2> $R/$O/bench.err; echo rc=$? )
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); grep -v "at::native\|rocclr" $f | head -12 | cut -c1-170
