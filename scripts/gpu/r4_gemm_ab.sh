#!/bin/bash
# r4: A/B timings of the batched-prefill GEMM variants (every command bounded); $2.. = variant strings "VAR=val VAR=val"
set -u
O=$1; shift; mkdir -p $O
export TMPDIR=/tmp
for v in "$@"; do
  ( env GL3_PF_GEMM2_ALL=${GL3_PF_GEMM2_ALL:-1} $v timeout 300 python scripts/gemm_ab.py llama-3-8b 4 2>&1 | tail -1 ) >> $O/ab.log 2>&1
done
cat $O/ab.log
