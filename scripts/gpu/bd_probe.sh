#!/bin/bash
set -u
O=${1:-gpurun_out/bd_probe}; mkdir -p $O
( cd scripts/probes; timeout 300 ./bd_probe 32 ) > $O/probe.log 2>&1
cat $O/probe.log
( timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_decode.py tests/test_gpu_prefill.py -m gpu -x -q 2>&1 | tail -8 ) > $O/pytest.log 2>&1
cat $O/pytest.log
( timeout 300 python scripts/bd_only.py qwen3-4b 32 16 2>&1 | tail -1 ) > $O/bd.log 2>&1
cat $O/bd.log
