#!/bin/bash
set -u
O=${1:-gpurun_out/q3b}; mkdir -p $O
( timeout 600 python bench.py --model qwen3-4b > $O/bench_qwen3_4b.json 2> $O/bench_q3.err )
( timeout 600 python bench.py --model qwen3-4b --decode-batch 32 > $O/bench_qwen3_4b_bd32.json 2> $O/bench_bd.err )
for f in $O/bench_*.json; do python - $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], d.get("value"), [(r.get("batch"), r.get("tok_s")) for r in d.get("pp_rows", [])], d.get("roofline",{}).get("frac"))
PY
done
