#!/bin/bash
set -u
O=gpurun_out/${1:-r3tp}; mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_tp.py -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest_tp.log 2>&1
cat $O/pytest_tp.log | cut -c1-300
