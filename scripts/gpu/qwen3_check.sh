#!/bin/bash
set -u
O=${1:-gpurun_out/qwen3}; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_fullsize.py -m gpu -x -q -k "qwen3 or batched or long or context" 2>&1 | tail -4 ) > $O/pytest.log 2>&1; cat $O/pytest.log
( timeout 300 python scripts/tg_only.py qwen3-4b 128 2>&1 | tail -1 ) > $O/tg.log 2>&1; cat $O/tg.log
( timeout 300 python scripts/bd_only.py qwen3-4b 32 16 2>&1 | tail -1 ) > $O/bd.log 2>&1; cat $O/bd.log
