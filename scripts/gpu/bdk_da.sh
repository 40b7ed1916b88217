#!/bin/bash
# A/B of the static-batched decode GEMM forms (Qwen3-4B, B = 32): bdw vs the k-slice form at ring depths 4 / 8.
set -u
O=${1:-gpurun_out/bdk_da}; mkdir -p $O
run() { echo "== $*" ; ( env "$@" timeout 300 python scripts/bd_only.py qwen3-4b 32 16 2>&1 | tail -1 ); }
{
run GL3_BDK=0
run GL3_BDK=1 GL3_BDK_P=2 GL3_BDK_GU=0 GL3_BDK_DA=4
run GL3_BDK=1 GL3_BDK_P=3 GL3_BDK_GU=0 GL3_BDK_DA=4
run GL3_BDK=1 GL3_BDK_P=2 GL3_BDK_GU=0 GL3_BDK_DA=8
run GL3_BDK=1 GL3_BDK_P=3 GL3_BDK_GU=0 GL3_BDK_DA=8
run GL3_BDK=0
} > $O/ab.log 2>&1
cat $O/ab.log
