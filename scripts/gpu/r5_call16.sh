#!/bin/bash
# round 5, call 16: final form of the matrix-core product GEMM: parity subset, then the two f32-activation bench lines
set -u
O=gpurun_out/r5_call16; mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_decode.py tests/test_gpu_fullsize.py tests/test_gpu_tp.py -m gpu -x -q --timeout 900 -k "f32_activation or q4 or Q4 or vl or 8b_layer or prefill or batched" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -12 ) > $O/pytest.log 2>&1; echo "== pytest"; tail -4 $O/pytest.log
for spec in "llama-3-8b q4_0" "llama-3-8b q8_0_f32act"; do
  set -- $spec
  ( timeout 900 python bench.py --steps 3 --warmup 1 --model $1 --wtype $2 --no-cpu-baseline > $O/bench_$1_$2.json 2> $O/bench_$1_$2.err; echo "$1 $2 rc=$?" )
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_$1_$2.json")); print("$1 $2: tg", d["value"], "pp", [(r.get("batch"), r.get("tok_s", r.get("error"))) for r in d["pp_rows"]])
except Exception as e: print("no json", e)
PY
done
