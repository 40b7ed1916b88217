cd $GRAFT_REPO_ROOT/scripts/probes && hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_bcast lds_bcast_probe.hip 2>/dev/null && timeout 120 /tmp/lds_bcast
