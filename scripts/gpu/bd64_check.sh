#!/bin/bash
set -u
O=${1:-gpurun_out/bd64}; mkdir -p $O
( timeout 300 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_decode.py -m gpu -x -q -k "batched" 2>&1 | tail -3 ) > $O/pytest.log 2>&1; cat $O/pytest.log
( timeout 120 python scripts/bd_only.py qwen3-4b 64 8 2>&1 | tail -1 ) > $O/bd64.log 2>&1; cat $O/bd64.log
( GL3_BD_GEMM=0 timeout 120 python scripts/bd_only.py qwen3-4b 64 8 2>&1 | tail -1 ) > $O/bd64_tiled.log 2>&1; cat $O/bd64_tiled.log
