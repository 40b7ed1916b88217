#!/bin/bash
# K-split residual projections (4 / 8 / 16 wavefronts per group) on top of the fused RMS prologue: parity incl. TP, then us / layer
set -u
O=gpurun_out/${1:-r3q4d}; mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_decode.py tests/test_gpu_fullsize.py tests/test_gpu_tp.py -m gpu -x -q -k "f16_and_q4_0 or f32_activation or q4_0 or golden or tied or gguf" 2>&1 | tail -5 ) > $O/pytest.log 2>&1
cat $O/pytest.log
for vlq in 0 1; do
  for spec in "llama-3-8b 2 x" "llama-3-8b 8 f32act"; do
    set -- $spec
    echo "GL3_VLQ=$vlq $1 type=$2: $(GL3_VLQ=$vlq timeout 300 python scripts/tg_only.py $1 8 $2 128 $3 2>&1 | tail -1)"
  done
done
