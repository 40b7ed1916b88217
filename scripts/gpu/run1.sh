#!/bin/bash
# GPU session 1 (round 2): parity tests incl. the new full-size shapes, matvec probe with / without the L2 touch prefetch, bench.
set -u
O=gpurun_out/r2a; mkdir -p $O
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > $O/pytest.log 2>&1
( cd scripts/probes && for t in 0 4 8; do echo "== MB_TOUCH=$t"; MB_TOUCH=$t timeout 120 ./matvec_bench 512 2>&1 | grep -v "^$"; done ) > $O/matvec_touch.log 2>&1
( timeout 600 python bench.py --steps 3 > $O/bench_8b.json 2> $O/bench_8b.err )
( GL3_TOUCH=8 timeout 300 python bench.py --steps 3 --no-pp --no-cpu-baseline > $O/bench_8b_touch8.json 2> $O/bench_8b_touch8.err )
( GL3_NO_FUSED_ATTN=1 timeout 300 python bench.py --steps 3 --no-pp --no-cpu-baseline > $O/bench_8b_nofused.json 2> $O/bench_8b_nofused.err )
tail -5 $O/pytest.log; python - <<'PY'
import json
for f in ("bench_8b","bench_8b_touch8","bench_8b_nofused"):
    try:
        d=json.loads(open("gpurun_out/r2a/%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["value"], {k:v["avg_us"] for k,v in d["kernels_eager_events"].items()}, d.get("pp_rows"))
    except Exception as e: print(f, "ERR", e)
PY
