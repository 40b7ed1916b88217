cd $GRAFT_REPO_ROOT
bash scripts/gpu/round6_final.sh gpurun_out/r6b tests bench depth 2>&1 | tail -40
