#!/bin/bash
set -u
O=gpurun_out/r2i; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "b32" 2>&1 | tail -3 ) > $O/pytest.log 2>&1
tail -2 $O/pytest.log
for v in "" "GL3_BD1=1"; do
  ( env $v timeout 300 python bench.py --model qwen3-4b --decode-batch 32 --steps 2 --warmup 1 --n-gen 64 > $O/bench_b32_$v.json 2> $O/bench_b32_$v.err )
  python - "$O/bench_b32_$v.json" "$v" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("B32", sys.argv[2], d["value"], "tok/s", d["ms_per_batched_step"], "ms/step frac", d["roofline"]["frac"])
except Exception as e: print("ERR", sys.argv[2], e)
PY
done
R=$GRAFT_REPO_ROOT
( cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_b32 -o b32 -- python $R/bench.py --model qwen3-4b --decode-batch 32 --steps 1 --warmup 1 --n-gen 32 > $R/$O/prof_bench_b32.json 2> $R/$O/prof_bench_b32.err; echo rc=$? )
f=$(find $O/prof_b32 -name "*kernel_stats.csv" | head -1); grep -v "at::native\|rocclr" $f | head -9 | cut -c1-150
