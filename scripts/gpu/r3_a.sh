#!/bin/bash
# round 3, first call: the new shape tests on the unchanged kernels + bench.py self-spawn (2 ranks sharing the GPU) + baseline line
set -u
O=gpurun_out/r3a; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -8 ) > $O/pytest_fullsize.log 2>&1
cat $O/pytest_fullsize.log
( GL3_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 1 --warmup 1 --model llama-3.2-1b --no-cpu-baseline > $O/bench_tp2_share.json 2> $O/bench_tp2_share.err; echo "tp2 rc=$?" )
tail -c 600 $O/bench_tp2_share.json; tail -5 $O/bench_tp2_share.err
( timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_8b.json 2> $O/bench_8b.err; echo "8b rc=$?" )
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3a/bench_8b.json"))
print("tg", d["value"], "pp", d["pp"]["tok_s"], {k:v["avg_us"] for k,v in d["kernel_classes"].items()}, d["kernels_eager_events"]["attention"])
PY
