#!/bin/bash
# r4: parity of every batched-prefill test with the fused prefill attention, then pp512 A/B (fused vs three kernels) on 8 layers of 8B and 16 of 1B
set -u
O=$1; mkdir -p $O
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_gpu_decode.py tests/test_gpu_fullsize.py -m gpu -x -q -k "prefill" 2>&1 | tail -5 ) > $O/pytest_prefill.log 2>&1
cat $O/pytest_prefill.log
for v in "GL3_PF_FUSED_ATTN=1" "GL3_PF_FUSED_ATTN=0"; do
  ( env $v timeout 300 python scripts/pp_only.py llama-3-8b 8 8 2>&1 | tail -1 | sed "s/^/[$v] /" ) >> $O/pp.log 2>&1
  ( env $v timeout 300 python scripts/pp_only.py llama-3.2-1b 16 8 2>&1 | tail -1 | sed "s/^/[$v] 1b /" ) >> $O/pp.log 2>&1
  ( env $v timeout 300 python scripts/pp_only.py qwen3-4b 8 8 2>&1 | tail -1 | sed "s/^/[$v] qwen3 /" ) >> $O/pp.log 2>&1
done
cat $O/pp.log
