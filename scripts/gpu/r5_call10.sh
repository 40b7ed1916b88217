#!/bin/bash
set -u
O=gpurun_out/r5_call10; mkdir -p $O
export TMPDIR=/tmp GL3_TP_SPIN_LIMIT=1000000
for mask in 31 1 2 4 8 16; do
  echo "== mask $mask"; ( GL3_TP_FOLD=2 GL3_TP_FOLD_MASK=$mask timeout 200 python scripts/debug_tp_fold.py mid-llama 4 2>&1 | grep -v "^rank . token" | tail -6 )
done
