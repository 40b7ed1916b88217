#!/bin/bash
# static-batched decode / small-chunk prefill: parity tests + step time
set -u
O=${1:-gpurun_out/bd_check}; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_decode.py -m gpu -x -q -k "batched" 2>&1 | tail -5 ) > $O/pytest.log 2>&1
cat $O/pytest.log
( timeout 300 python scripts/bd_only.py qwen3-4b 32 16 2>&1 | tail -1 ) > $O/bd.log 2>&1; cat $O/bd.log
( GL3_NO_FUSED_QUANT=1 timeout 300 python scripts/bd_only.py qwen3-4b 32 16 2>&1 | tail -1 ) > $O/bd_unfused.log 2>&1; cat $O/bd_unfused.log
