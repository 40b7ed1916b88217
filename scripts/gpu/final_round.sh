#!/bin/bash
# Round-end measurement: smoke, headline bench, the other bench lines, rocprofv3 profiles.  Everything bounded by timeout.
set -u
O=${1:-gpurun_out/final}; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ) > $O/smoke.log 2>&1; cat $O/smoke.log
( timeout 900 python bench.py > $O/bench_8b.json 2> $O/bench_8b.err ); tail -c 600 $O/bench_8b.json | head -c 300; echo
( timeout 600 python bench.py --model llama-3.2-1b > $O/bench_llama32_1b.json 2> $O/bench_1b.err )
( timeout 600 python bench.py --model qwen3-4b > $O/bench_qwen3_4b.json 2> $O/bench_q3.err )
( timeout 600 python bench.py --model qwen3-4b --decode-batch 32 > $O/bench_qwen3_4b_bd32.json 2> $O/bench_bd.err )
( timeout 600 python bench.py --model llama-3.2-1b --wtype f16 --no-cpu-baseline > $O/bench_llama32_1b_f16.json 2> $O/bench_f16.err )
( timeout 600 python bench.py --wtype q4_0 --no-cpu-baseline > $O/bench_8b_q4_0.json 2> $O/bench_q40.err )
for f in $O/bench_*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d.get("value"), d.get("unit"), "pp", [(r.get("batch"), r.get("tok_s")) for r in d.get("pp_rows", [])], "frac", d.get("roofline",{}).get("frac"))
except Exception as e: print("ERR", sys.argv[1], e)
PY
done
bash scripts/gpu/profile_round.sh $O/prof
