#!/bin/bash
# round 5, call 2: what about the UNCACHED arena makes in-process ranks compute wrong rows?  (transport checksums were clean)
set -u
O=gpurun_out/r5_call2; mkdir -p $O
export TMPDIR=/tmp
run() { tag=$1; shift; ( env "$@" DBG_ITERS=10 timeout 300 python scripts/debug_tp_vl.py ) > $O/$tag.log 2>&1; echo "== $tag: $(grep SUMMARY $O/$tag.log)"; grep -m3 -A3 "first bad\|second read" $O/$tag.log | grep -v "^--" | head -8; }
run plain X=1
run barrier DBG_BARRIER=1
run cached GL3_TP_ARENA=cached
run finegrained GL3_TP_ARENA=finegrained
run pad GL3_TP_ARENA_PAD=1
run lock GL3_TP_ARENA_LOCK=1
run leak GL3_TP_ARENA_LEAK=1
run nosdma HSA_ENABLE_SDMA=0
run f16 DBG_ORDER=f16,int8
