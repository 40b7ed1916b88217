#!/bin/bash
# fused RMSNorm prologue of the vector-order matvecs: parity, then us / layer (8 layers, 8B shapes; 1B F16) with and without it
set -u
O=gpurun_out/${1:-r3q4c}; mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_decode.py tests/test_gpu_fullsize.py tests/test_gpu_tp.py -m gpu -x -q -k "f16_and_q4_0 or f32_activation or q4_0 or golden or tied or gguf" 2>&1 | tail -5 ) > $O/pytest.log 2>&1
cat $O/pytest.log
for rms in 0 1; do
  for spec in "llama-3-8b 2 x" "llama-3-8b 8 f32act" "llama-3.2-1b 1 x"; do
    set -- $spec
    echo "GL3_VL_RMS=$rms GL3_VLQ=0 $1 type=$2: $(GL3_VL_RMS=$rms GL3_VLQ=0 timeout 300 python scripts/tg_only.py $1 8 $2 128 $3 2>&1 | tail -1)"
  done
done
