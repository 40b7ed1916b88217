#!/bin/bash
# round 5, call 8: persistent scores kernel: parity, then depth A/B
set -u
O=gpurun_out/r5_call8; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_decode.py -m gpu -x -q --timeout 600 2>&1 | tail -6 ) > $O/pytest.log 2>&1; echo "== pytest"; tail -3 $O/pytest.log
for mode in 1 0; do
( GL3_ATTN_SCORES_LOOP=$mode timeout 600 python bench.py --steps 3 --warmup 1 --depth 1024,4096,16384 --no-cpu-baseline --no-pp 2> $O/bench_depth_$mode.err | tail -1 ) > $O/bench_depth_$mode.json; echo "== bench depth GL3_ATTN_SCORES_LOOP=$mode"; python - <<PY
import json
try:
    d = json.load(open("$O/bench_depth_$mode.json"))
    print("tg128", d["value"])
    for r in d.get("depth_rows", []): print({k: r[k] for k in ("test", "tok_s", "attention_us_per_layer") if k in r} or r)
except Exception as e:
    print("no json:", e); print(open("$O/bench_depth_$mode.err").read()[-2000:])
PY
done
DEPTHS="4096 16384" bash scripts/gpu/r5_prof_depth.sh 2>&1 | grep "gl3::attn\|== depth"
