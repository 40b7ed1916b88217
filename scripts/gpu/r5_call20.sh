#!/bin/bash
# round 5, call 20: after the last kernel change — the whole GPU suite in one process, kernel statistics and the FETCH_SIZE pass of the headline command
set -u
O=gpurun_out/r5_final2; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -E "passed|failed|^FAILED|rror" | tail -4 ) > $O/pytest_all.log 2>&1; echo "== suite: $(tail -1 $O/pytest_all.log)"
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o k -- python $R/bench.py --steps 2 --no-cpu-baseline > $R/$O/bench_traced.json 2> $R/$O/bench_traced.err; echo trace rc=$? )
( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_fetch -o f -- python $R/bench.py --steps 1 --no-pp --no-cpu-baseline > $R/$O/bench_pmc.json 2> $R/$O/bench_pmc.err; echo pmc rc=$? )
find $O -name "*kernel_trace.csv" -delete
( timeout 600 python bench.py --steps 3 --warmup 1 --depth 256,512,1024,4096,16384 --no-pp > $O/bench_8b_depth.json 2> $O/bench_8b_depth.err; echo "depth rc=$?" )
DEPTHS="4096 16384" bash scripts/gpu/r5_prof_depth.sh 2>&1 | grep "gl3::attn\|== depth"
du -sh $O
