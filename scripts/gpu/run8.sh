#!/bin/bash
set -u
O=gpurun_out/r2g; mkdir -p $O
export TMPDIR=/tmp
( cd scripts/probes && timeout 120 ./matvec_bench 448 512 299 2>&1 | grep -v "^$\|exact-sum\|WG0" | head -16 ) > $O/probes.log 2>&1
cat $O/probes.log
R=$GRAFT_REPO_ROOT
( cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_b32 -o b32 -- python $R/bench.py --model qwen3-4b --decode-batch 32 --steps 1 --warmup 1 --n-gen 32 > $R/$O/bench_b32.json 2> $R/$O/bench_b32.err; echo rc=$? )
f=$(find $O/prof_b32 -name "*kernel_stats.csv" | head -1); echo $f; head -16 $f | cut -c1-160
tail -2 $O/bench_b32.json | cut -c1-400
