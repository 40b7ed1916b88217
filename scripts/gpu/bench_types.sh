#!/bin/bash
# bench lines of the f32-activation weight types (tg128 + pp512): 8B Q4_0, 8B Q8_0-f32act, 1B F16
set -u
O=gpurun_out/${1:-r3types}; mkdir -p $O
export TMPDIR=/tmp
for spec in "llama-3-8b q4_0" "llama-3-8b q8_0_f32act" "llama-3.2-1b f16"; do
  set -- $spec
  ( timeout 900 python bench.py --steps 3 --warmup 1 --model $1 --wtype $2 --no-cpu-baseline > $O/bench_$1_$2.json 2> $O/bench_$1_$2.err; echo "$1 $2 rc=$?" )
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_$1_$2.json")); print("$1 $2: tg", d["value"], "pp", [(r.get("batch"), r.get("tok_s", r.get("error"))) for r in d["pp_rows"]], "roofline", d["roofline"]["frac"], d["roofline"]["avg_us"], {k:v["avg_us"] for k,v in d["kernels_eager_events"].items()})
except Exception as e: print("no json", e)
PY
done
