#!/bin/bash
set -u
O=gpurun_out/r2f; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_fullsize.py tests/test_gpu_seqsum.py -m gpu -x -q 2>&1 | tail -8 ) > $O/pytest.log 2>&1
tail -4 $O/pytest.log
( cd scripts/probes && timeout 120 ./matvec_bench 512 2>&1 | grep -v "^$\|exact-sum" | head -12 ) > $O/probes.log 2>&1
cat $O/probes.log
( timeout 300 python bench.py --steps 3 --no-pp --no-cpu-baseline > $O/bench_8b.json 2> $O/bench_8b.err )
python - <<'PY'
import json
for f in ("bench_8b",):
    try:
        d=json.loads(open("gpurun_out/r2f/%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["value"], {k:v["avg_us"] for k,v in d["kernels_eager_events"].items()}, {k:v["avg_us"] for k,v in d["kernel_classes"].items()})
    except Exception as e: print(f, "ERR", e); print(open("gpurun_out/r2f/%s.err"%f).read()[-800:])
PY
