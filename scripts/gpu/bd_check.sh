#!/bin/bash
# static-batched decode: parity tests + kernel statistics
set -u
O=${1:-gpurun_out/bd_check}; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_decode.py -m gpu -x -q -k "batched_decode" 2>&1 | tail -5 ) > $O/pytest.log 2>&1
cat $O/pytest.log
bash scripts/gpu/prof_bd.sh $O
