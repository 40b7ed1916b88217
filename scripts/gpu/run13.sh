#!/bin/bash
set -u
O=gpurun_out/r2l; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_decode.py -m gpu -x -q -k "phi3 or granite" 2>&1 | tail -8 ) > $O/pytest.log 2>&1
tail -4 $O/pytest.log
( timeout 600 python -m pytest tests/test_gpu_tp.py -m gpu -x -q -k "phi3" 2>&1 | tail -6 ) > $O/pytest_tp.log 2>&1
tail -4 $O/pytest_tp.log
