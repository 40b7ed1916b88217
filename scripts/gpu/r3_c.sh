#!/bin/bash
# head_size 96, K-quant conversion, sampler, B = 32 batched decode with the two-stage argmax
set -u
O=gpurun_out/${1:-r3c}; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_kquant.py tests/test_gpu_sampling.py tests/test_gpu_fullsize.py -m gpu -x -q -k "hs96 or kquant or sampl or static_batched or b32" 2>&1 | tail -6 ) > $O/pytest.log 2>&1
cat $O/pytest.log
( timeout 600 python bench.py --model qwen3-4b --decode-batch 32 --steps 2 --warmup 1 > $O/bench_bd32.json 2> $O/bench_bd32.err; echo "bd32 rc=$?" )
python - <<PY
import json
d=json.load(open("$O/bench_bd32.json")); print("bd32", d["value"], d["ms_per_batched_step"], d["roofline"]["frac"])
PY
