#!/bin/bash
# fused attention + wo launch: decode parity, then the bench line (with and without the fusion)
set -u
O=gpurun_out/${1:-r3d}; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_fullsize.py -m gpu -x -q -k "decode_matches or golden or handover or graph_and_eager or sequential_prefill or prefill512 or 1b_shaped" 2>&1 | tail -8 ) > $O/pytest.log 2>&1
cat $O/pytest.log
for f in 1 0; do
( GL3_FUSE_ATTN_WO=$f timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pp > $O/bench_8b_f$f.json 2> $O/bench_8b_f$f.err; echo "8b fuse=$f rc=$?" )
python - <<PY
import json
d=json.load(open("$O/bench_8b_f$f.json"))
print("fuse=$f tg", d["value"], {k:v["avg_us"] for k,v in d["kernels_eager_events"].items()})
PY
done
