#!/bin/bash
# Q4_0 / Q8_0-f32act decode variants: parity first, then us / layer of 8 layers at the 8B shapes for GL3_VLQ = 0 / 1
set -u
O=gpurun_out/${1:-r3q4b}; mkdir -p $O
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_gpu_decode.py tests/test_gpu_fullsize.py -m gpu -x -q -k "f16_and_q4_0 or f32_activation or q4_0 or golden" 2>&1 | tail -5 ) > $O/pytest.log 2>&1
cat $O/pytest.log
( GL3_VLQ=0 timeout 600 python -m pytest tests/test_gpu_decode.py -m gpu -x -q -k "f16_and_q4_0 or q8_0_with_f32" 2>&1 | tail -3 )
for v in 0 1; do
  for spec in "2 x" "8 f32act"; do
    set -- $spec
    echo "GL3_VLQ=$v type=$1: $(GL3_VLQ=$v timeout 300 python scripts/tg_only.py llama-3-8b 8 $1 128 $2 2>&1 | tail -1)"
  done
done
