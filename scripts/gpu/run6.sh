#!/bin/bash
set -u
O=gpurun_out/r2e; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_decode.py -m gpu -x -q -k "f16 or q4_0 or golden" 2>&1 | tail -25 ) > $O/pytest.log 2>&1
tail -14 $O/pytest.log
( timeout 300 python bench.py --model llama-3.2-1b --wtype f16 --steps 3 --no-cpu-baseline --no-pp > $O/bench_1b_f16.json 2> $O/bench_1b_f16.err )
( timeout 400 python bench.py --model llama-3-8b --wtype q4_0 --steps 3 --no-cpu-baseline --no-pp > $O/bench_8b_q4_0.json 2> $O/bench_8b_q4_0.err )
( timeout 400 python bench.py --model llama-3.2-1b --wtype q4_0 --steps 3 --no-cpu-baseline --no-pp > $O/bench_1b_q4_0.json 2> $O/bench_1b_q4_0.err )
python - <<'PY'
import json
for f in ("bench_1b_f16","bench_8b_q4_0","bench_1b_q4_0"):
    try:
        d=json.loads(open("gpurun_out/r2e/%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["value"], "token-frac", d["token_level"]["frac_of_hbm_peak"], {k:(v["avg_us"], v.get("frac_of_hbm_peak")) for k,v in d["kernel_classes"].items()})
    except Exception as e: print(f, "ERR", e); print(open("gpurun_out/r2e/%s.err"%f).read()[-600:])
PY
