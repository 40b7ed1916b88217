cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "behind_1000 or prefill512" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_decode.py -m gpu -x -q -k "long_context or beyond_20k or chunks_above" 2>&1 | tail -3
GL3_PF_FUSED_ATTN=0 timeout 600 python -m pytest tests/test_gpu_decode.py -m gpu -x -q -k "long_context or chunks_above or batched_prefill" 2>&1 | tail -3
DEPTH=4096 bash scripts/gpu/pp_depth_prof.sh gpurun_out/ppd4 2>&1 | grep "pf_\|pp512"
