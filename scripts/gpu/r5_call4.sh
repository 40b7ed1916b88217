#!/bin/bash
# round 5, call 4: the whole GPU suite (TP types re-enabled, new gl3_bench test) + decode at depth
set -u
O=gpurun_out/r5_call4; mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q --timeout 600 2>&1 | tail -15 ) > $O/pytest.log 2>&1; echo "== pytest"; cat $O/pytest.log
( timeout 900 python bench.py --steps 3 --warmup 1 --depth 4096,16384 --no-cpu-baseline 2> $O/bench_depth.err | tail -1 ) > $O/bench_depth.json; echo "== bench depth"; python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r5_call4/bench_depth.json"))
    print("tg128", d["value"], "pp", d["pp"]["tok_s"] if d.get("pp") else None)
    for r in d.get("depth_rows", []): print(r)
except Exception as e:
    print("no json:", e); print(open("gpurun_out/r5_call4/bench_depth.err").read()[-2000:])
PY
