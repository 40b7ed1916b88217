#!/bin/bash
# round 5, call 5: the split long-context decode attention (attn_softmax_kernel + attn_pv_kernel): parity, then tg at depth old vs new
set -u
O=gpurun_out/r5_call5; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_fullsize.py tests/test_gpu_tp.py -m gpu -x -q --timeout 600 2>&1 | tail -8 ) > $O/pytest.log 2>&1; echo "== pytest"; cat $O/pytest.log
for mode in 1 0; do
( GL3_ATTN_LONG=$mode timeout 600 python bench.py --steps 3 --warmup 1 --depth 256,1024,4096,16384 --no-cpu-baseline --no-pp 2> $O/bench_depth_$mode.err | tail -1 ) > $O/bench_depth_$mode.json; echo "== bench depth GL3_ATTN_LONG=$mode"; python - <<PY
import json
try:
    d = json.load(open("$O/bench_depth_$mode.json"))
    print("tg128", d["value"])
    for r in d.get("depth_rows", []): print({k: r[k] for k in ("test", "tok_s", "attention_us_per_layer", "kv_read_us_per_layer_at_hbm_peak") if k in r} or r)
except Exception as e:
    print("no json:", e); print(open("$O/bench_depth_$mode.err").read()[-2000:])
PY
done
