#!/bin/bash
# the whole GPU suite in ONE process, the way the driver runs it at round end
set -u
O=gpurun_out/${1:-full_suite}; mkdir -p $O
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 ) > $O/pytest_all.log 2>&1
grep -n "passed\|failed\|Fatal\|rror" $O/pytest_all.log | head -10; tail -3 $O/pytest_all.log
