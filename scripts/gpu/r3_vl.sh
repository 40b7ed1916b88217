#!/bin/bash
# batched prefill / static-batched decode of the f32-activation weight types: parity, then pp512 / tg128 of the affected models
set -u
O=gpurun_out/${1:-r3vl}; mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_decode.py tests/test_gpu_fullsize.py -m gpu -x -q -k "f32_activation or f16_and_q4_0 or error_behaviour or 8b_layer_shape" 2>&1 | tail -15 ) > $O/pytest.log 2>&1
cat $O/pytest.log
if [ "${2:-}" = "bench" ]; then
for spec in "llama-3.2-1b f16" "llama-3-8b q4_0" "llama-3-8b q8_0_f32act"; do
  set -- $spec
  ( timeout 900 python bench.py --steps 2 --warmup 1 --model $1 --wtype $2 --no-cpu-baseline > $O/bench_$1_$2.json 2> $O/bench_$1_$2.err; echo "$1 $2 rc=$?" )
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_$1_$2.json")); print("$1 $2: tg", d["value"], "pp", [(r.get("batch"), r.get("tok_s", r.get("error"))) for r in d["pp_rows"]])
except Exception as e: print("no json", e)
PY
done
fi
