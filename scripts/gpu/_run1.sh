cd $GRAFT_REPO_ROOT
GL3_LIB=$GRAFT_REPO_ROOT/gpullama3.java_amd/libgpullama_hip_fat.so python scripts/pp_only.py llama-3-8b 1 8 > gpurun_out/fat3.log 2>&1
