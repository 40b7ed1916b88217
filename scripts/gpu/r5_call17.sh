#!/bin/bash
# round 5, call 17: where the four-launch attention should take over from the two-launch pair (GL3_ATTN_MID)
set -u
O=gpurun_out/r5_call17; mkdir -p $O
export TMPDIR=/tmp
for mid in 768 256; do
( GL3_ATTN_MID=$mid timeout 600 python bench.py --steps 2 --warmup 1 --depth 256,384,512,640 --no-cpu-baseline --no-pp 2> $O/bench_$mid.err | tail -1 ) > $O/bench_$mid.json; echo "== GL3_ATTN_MID=$mid"; python - <<PY
import json
try:
    d = json.load(open("$O/bench_$mid.json"))
    for r in d.get("depth_rows", []): print(r["test"], r.get("tok_s"), r.get("attention_us_per_layer"))
except Exception as e:
    print("no json:", e); print(open("$O/bench_$mid.err").read()[-1500:])
PY
done
