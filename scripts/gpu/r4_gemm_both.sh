#!/bin/bash
# r4: timing stamps (diagnostic build) + A/B lines.  $1 = out dir, rest = variant strings
set -u
O=$1; shift; mkdir -p $O
bash scripts/gpu/r4_gemm_timing.sh $O 2>&1 | head -3
bash scripts/gpu/r4_gemm_ab.sh $O "$@"
