#!/bin/bash
# matvec rewrite: parity (decode + fullsize suites) then the bench line
set -u
O=gpurun_out/${1:-r3b}; mkdir -p $O
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_gpu_decode.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest.log 2>&1
cat $O/pytest.log
( timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_8b.json 2> $O/bench_8b.err; echo "8b rc=$?" )
tail -3 $O/bench_8b.err
python - <<PY
import json
d=json.load(open("$O/bench_8b.json"))
print("tg", d["value"], "pp", d["pp"]["tok_s"], {k:(v["avg_us"], v["frac_of_hbm_peak"]) for k,v in d["kernel_classes"].items()}, d["kernels_eager_events"]["attention"]["avg_us"])
PY
