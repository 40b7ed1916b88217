#!/bin/bash
# K-split decode matvec of Q4_0 / Q8_0-f32act: parity (decode, prefill hand-over, TP slices), then tg128 of 8B Q4_0 / Q8_0-f32act
set -u
O=gpurun_out/${1:-r3q4}; mkdir -p $O
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_gpu_decode.py tests/test_gpu_fullsize.py tests/test_gpu_tp.py -m gpu -x -q -k "f16_and_q4_0 or f32_activation or q4_0 or golden" 2>&1 | tail -8 ) > $O/pytest.log 2>&1
cat $O/pytest.log
for spec in "llama-3-8b q4_0" "llama-3-8b q8_0_f32act" ${Q4_EXTRA:-}; do
  set -- $spec
  ( timeout 600 python bench.py --steps 2 --warmup 1 --model $1 --wtype $2 --no-cpu-baseline > $O/bench_$1_$2.json 2> $O/bench_$1_$2.err; echo "$1 $2 rc=$?" )
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_$1_$2.json")); print("$1 $2: tg", d["value"], "pp", [(r.get("batch"), r.get("tok_s", r.get("error"))) for r in d["pp_rows"]], {k:v["avg_us"] for k,v in d["kernels_eager_events"].items()})
except Exception as e: print("no json", e)
PY
done
