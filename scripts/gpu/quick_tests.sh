#!/bin/bash
# GPU suite without the (slow, in-process multi-rank) tensor-parallel file
set -u
O=${1:-gpurun_out/quick_tests}; mkdir -p $O
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_gpu_decode.py tests/test_gpu_fullsize.py tests/test_gpu_kquant.py tests/test_gpu_sampling.py tests/test_gpu_seqsum.py -m gpu -x -q 2>&1 | tail -5 ) > $O/pytest.log 2>&1
cat $O/pytest.log
