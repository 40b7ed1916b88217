#!/bin/bash
set -u
O=gpurun_out/r2c; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_decode.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -8 ) > $O/pytest.log 2>&1
( cd scripts/probes && timeout 60 ./seqsum_time; timeout 120 ./matvec_bench 512 2>&1 | grep -v "^$" ) > $O/probes.log 2>&1
( timeout 300 python bench.py --steps 3 --no-pp --no-cpu-baseline > $O/bench_8b.json 2> $O/bench_8b.err )
( GL3_NO_FUSED_ATTN=1 timeout 300 python bench.py --steps 3 --no-pp --no-cpu-baseline > $O/bench_8b_nofused.json 2> $O/bench_8b_nofused.err )
tail -3 $O/pytest.log; head -40 $O/probes.log; python - <<'PY'
import json
for f in ("bench_8b","bench_8b_nofused"):
    try:
        d=json.loads(open("gpurun_out/r2c/%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["value"], {k:v["avg_us"] for k,v in d["kernels_eager_events"].items()})
    except Exception as e: print(f, "ERR", e)
PY
