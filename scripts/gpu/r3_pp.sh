#!/bin/bash
# prefill GEMM change: parity of the batched int8 path, then the bench line (pp512 + per-GEMM times)
set -u
O=gpurun_out/${1:-r3pp}; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_fullsize.py -m gpu -x -q -k "batched_prefill_is_bit or prefill512 or long_context or b32" 2>&1 | tail -6 ) > $O/pytest.log 2>&1
cat $O/pytest.log
( timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_8b.json 2> $O/bench_8b.err; echo "8b rc=$?" )
python - <<PY
import json
d=json.load(open("$O/bench_8b.json"))
print("tg", d["value"], "pp", [(r["batch"], r.get("tok_s")) for r in d["pp_rows"]], {k:(v["avg_us"], v["frac_of_int8_mfma_peak"]) for k,v in d["roofline_pp"]["gemms"].items()})
PY
