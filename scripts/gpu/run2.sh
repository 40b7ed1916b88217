#!/bin/bash
set -u
O=gpurun_out/r2b; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o tg -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --no-pp --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench.json 2> $GRAFT_REPO_ROOT/$O/bench.err
cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats*" | head; f=$(find $O/prof -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-200
