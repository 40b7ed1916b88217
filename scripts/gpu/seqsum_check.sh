#!/bin/bash
set -u
O=${1:-gpurun_out/seqsum}; mkdir -p $O
( cd scripts/probes && timeout 60 ./seqsum_time 2>&1 | tail -6 ) > $O/probe.log 2>&1; cat $O/probe.log
( timeout 600 python -m pytest tests/test_gpu_seqsum.py tests/test_gpu_sampling.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3 ) > $O/pytest.log 2>&1; cat $O/pytest.log
( timeout 300 python scripts/tg_only.py llama-3-8b 128 2>&1 | tail -1 ) > $O/tg.log 2>&1; cat $O/tg.log
( timeout 300 python scripts/bd_only.py qwen3-4b 32 16 2>&1 | tail -1 ) > $O/bd.log 2>&1; cat $O/bd.log
