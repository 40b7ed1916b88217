#!/bin/bash
set -u
O=gpurun_out/r2d; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_tp.py -m gpu -q 2>&1 | tail -30 ) > $O/pytest_tp.log 2>&1
tail -12 $O/pytest_tp.log
