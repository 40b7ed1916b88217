#!/bin/bash
# r4: parity of the batched prefill with the new GEMM + A/B timings of its variants (every command bounded)
set -u
O=${1:-gpurun_out/r4_gemm}; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_decode.py -m gpu -x -q -k "batched_prefill_is_bit_identical or long_context" 2>&1 | tail -5 ) > $O/pytest_prefill.log 2>&1
cat $O/pytest_prefill.log
( timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "prefill512" 2>&1 | tail -5 ) > $O/pytest_full.log 2>&1
cat $O/pytest_full.log
for v in "GL3_PF_GEMM2=0" "GL3_PF_GEMM2=1" "GL3_PF_GEMM2=2" "GL3_PF_GEMM2=2 GL3_PF_GEMM2_OCC=3" "GL3_PF_GEMM2=1 GL3_PF_GEMM2_OCC=3"; do
  ( env $v timeout 300 python scripts/gemm_ab.py llama-3-8b 4 2>&1 | tail -1 ) >> $O/ab.log 2>&1
done
cat $O/ab.log
