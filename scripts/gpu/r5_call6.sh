#!/bin/bash
# round 5, call 6: long-context decode attention v5 (scores+tilemax, exp, sum, pv): parity, kernel stats at depth, bench at depth
set -u
O=gpurun_out/r5_call6; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_fullsize.py -m gpu -x -q --timeout 600 2>&1 | tail -8 ) > $O/pytest.log 2>&1; echo "== pytest"; tail -4 $O/pytest.log
DEPTHS="512 4096 16384" bash scripts/gpu/r5_prof_depth.sh 2>&1 | grep -v "^W2026\|pf_\|at::native\|rocclr"
( timeout 600 python bench.py --steps 3 --warmup 1 --depth 256,512,1024,4096,16384 --no-cpu-baseline --no-pp 2> $O/bench_depth.err | tail -1 ) > $O/bench_depth.json; echo "== bench depth"; python - <<PY
import json
try:
    d = json.load(open("$O/bench_depth.json"))
    print("tg128", d["value"])
    for r in d.get("depth_rows", []): print({k: r[k] for k in ("test", "tok_s", "attention_us_per_layer", "kv_read_us_per_layer_at_hbm_peak") if k in r} or r)
except Exception as e:
    print("no json:", e); print(open("$O/bench_depth.err").read()[-2000:])
PY
