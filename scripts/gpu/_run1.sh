cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_decode.py -m gpu -x -q -k "prefill or chunks or long_context" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "behind_1000 or prefill512" 2>&1 | tail -3
for rep in 1 2; do for v in "GL3_NOOP=1" "GL3_PF_FUSED_MFMA=0"; do echo "== $v"; env $v python scripts/pp_only.py llama-3-8b 8 8 2>&1 | grep pp512; done; done
for v in "GL3_NOOP=1" "GL3_PF_FUSED_MFMA=0"; do echo "== $v qwen3-4b"; env $v python scripts/pp_only.py qwen3-4b 8 8 2>&1 | grep pp512; echo "== $v 1b";  env $v python scripts/pp_only.py llama-3.2-1b 8 8 2>&1 | grep pp512; done
