#!/bin/bash
set -u
O=${1:-gpurun_out/attn}; mkdir -p $O
( cd scripts/probes && timeout 60 ./attn_head_probe 0 ) > $O/probe.log 2>&1; cat $O/probe.log
( timeout 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_fullsize.py -m gpu -x -q -k "c_oracle_live or golden or batched_decode or layer_prefill" 2>&1 | tail -4 ) > $O/pytest.log 2>&1; cat $O/pytest.log
( timeout 300 python scripts/tg_only.py llama-3-8b 128 2>&1 | tail -1 ) > $O/tg.log 2>&1; cat $O/tg.log
