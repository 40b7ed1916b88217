#!/bin/bash
# round 5, call 3: the arena pool (fix) looped in both layouts; the unpooled arena as the control; the cached <-> uncached recycling probe
set -u
O=gpurun_out/r5_call3; mkdir -p $O
export TMPDIR=/tmp
( timeout 200 scripts/probes/uncached_recycle_probe ) > $O/recycle_probe.log 2>&1; echo "== recycle probe"; cat $O/recycle_probe.log
run() { tag=$1; shift; ( env "$@" timeout 900 python scripts/debug_tp_vl.py ) > $O/$tag.log 2>&1; echo "== $tag: $(grep SUMMARY $O/$tag.log)"; grep -m2 -A2 "first bad" $O/$tag.log | head -6; }
run pooled DBG_ITERS=50
run pooled_dbg DBG_ITERS=25 GL3_TP_DEBUG=1
run unpooled DBG_ITERS=10 GL3_TP_ARENA=unpooled
( timeout 900 python scripts/debug_tp_procs.py 4 2 50 ) > $O/procs_q4.log 2>&1; echo "== procs q4 x4 x50"; grep -c "differ" $O/procs_q4.log; tail -1 $O/procs_q4.log
( timeout 900 python scripts/debug_tp_procs.py 2 8 50 f32act ) > $O/procs_f32.log 2>&1; echo "== procs f32act x2 x50"; tail -1 $O/procs_f32.log
( timeout 900 python scripts/debug_tp_procs.py 4 2 20 ) > $O/procs_q4_b.log 2>&1; echo "== procs q4 x4 x20 (again)"; tail -1 $O/procs_q4_b.log
