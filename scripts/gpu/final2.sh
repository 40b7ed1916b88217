#!/bin/bash
# full GPU suite + refreshed headline / B=32 bench lines
set -u
O=${1:-gpurun_out/final2}; mkdir -p $O
export TMPDIR=/tmp
bash scripts/gpu/full_tests.sh $O
( timeout 900 python bench.py > $O/bench_8b.json 2> $O/bench_8b.err )
( timeout 600 python bench.py --model qwen3-4b --decode-batch 32 > $O/bench_qwen3_4b_bd32.json 2> $O/bench_bd.err )
for f in $O/bench_*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d.get("value"), d.get("unit"), "pp", [(r.get("batch"), r.get("tok_s")) for r in d.get("pp_rows", [])], "frac", d.get("roofline",{}).get("frac"), d.get("roofline",{}).get("traffic_source"))
except Exception as e: print("ERR", sys.argv[1], e)
PY
done
