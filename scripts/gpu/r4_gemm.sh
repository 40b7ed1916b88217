#!/bin/bash
# r4: parity of the batched prefill (whatever GL3_PF_GEMM2 the caller exports), then A/B timings of the variants given as "VAR=val ..." strings
set -u
O=$1; shift; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_fullsize.py -m gpu -x -q -k "batched_prefill_is_bit_identical or long_context or prefill512" 2>&1 | tail -5 ) > $O/pytest_prefill.log 2>&1
cat $O/pytest_prefill.log
bash scripts/gpu/r4_gemm_ab.sh $O "$@"
