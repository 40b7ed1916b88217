#!/bin/bash
# the whole GPU suite, file by file (bounded), summary lines into $1/pytest_*.log
set -u
O=${1:-gpurun_out/full_tests}; mkdir -p $O
export TMPDIR=/tmp
for f in tp decode fullsize kquant sampling seqsum run_host; do
  ( timeout 1200 python -m pytest tests/test_gpu_$f.py -m gpu -x -q --timeout 300 2>&1 | tail -6 ) > $O/pytest_$f.log 2>&1
  echo "== $f"; tail -3 $O/pytest_$f.log
done
