cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "behind_1000 or prefill512" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_decode.py -m gpu -x -q -k "prefill or chunks or long_context or beyond_20k" 2>&1 | tail -3
for rep in 1 2; do PP_DEPTH=4096 python scripts/pp_only.py llama-3-8b 4 8 2>&1 | grep pp512; python scripts/pp_only.py llama-3-8b 8 8 2>&1 | grep pp512; done
python scripts/pp_only.py qwen3-4b 8 8 2>&1 | grep pp512; python scripts/pp_only.py llama-3.2-1b 8 8 2>&1 | grep pp512
DEPTH=4096 bash scripts/gpu/pp_depth_prof.sh gpurun_out/ppd5 2>&1 | grep "pf_scores\|pf_pv\|pf_soft\|pp512"
