#!/bin/bash
set -u
O=gpurun_out/r2j; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_sampling.py tests/test_gpu_seqsum.py -m gpu -x -q 2>&1 | tail -25 ) > $O/pytest.log 2>&1
tail -25 $O/pytest.log
( cd scripts/probes && timeout 60 ./seqsum_time 2>&1 | tail -4; timeout 120 ./matvec_bench 256 512 2>&1 | grep -v "^$\|exact-sum\|WG0" | head -10 ) > $O/probes.log 2>&1
cat $O/probes.log
