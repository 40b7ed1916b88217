#!/bin/bash
set -u
O=${1:-gpurun_out/f32act}; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_decode.py -m gpu -x -q -k "f32_activation or golden or f16_and_q4_0" 2>&1 | tail -5 ) > $O/pytest.log 2>&1; cat $O/pytest.log
