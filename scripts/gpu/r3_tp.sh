#!/bin/bash
set -u
O=gpurun_out/${1:-r3tp}; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_tp.py -m gpu -x -q --durations=12 -k "mid-llama-2-8 or mid-llama-8-8 or static_batched_decode_under or between_processes[mid-llama-2-8]" 2>&1 | tail -25 ) > $O/pytest_tp.log 2>&1
cat $O/pytest_tp.log | cut -c1-200
