#!/bin/bash
set -u
O=gpurun_out/r2k; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_kquant.py tests/test_gpu_sampling.py -m gpu -x -q 2>&1 | tail -12 ) > $O/pytest.log 2>&1
tail -6 $O/pytest.log
( timeout 600 python -m pytest tests/test_gpu_tp.py -m gpu -x -q -k "granite or mid-llama-2-8 or static" --durations=5 2>&1 | tail -12 ) > $O/pytest_tp.log 2>&1
tail -10 $O/pytest_tp.log
