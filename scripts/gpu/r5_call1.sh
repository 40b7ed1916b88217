#!/bin/bash
# round 5, first GPU call: any-order launch probe; the in-process TP stale-row flake (plain, checksummed gather, cached arena) looped;
# the one-process-per-rank layout looped; eager vs graph decode
set -u
O=gpurun_out/r5_call1; mkdir -p $O
export TMPDIR=/tmp
( timeout 120 scripts/probes/anyorder_probe ) > $O/anyorder.log 2>&1; echo "== anyorder"; cat $O/anyorder.log
( DBG_ITERS=12 timeout 600 python scripts/debug_tp_vl.py ) > $O/tp_inproc_plain.log 2>&1; echo "== inproc plain"; grep -v "^\[" $O/tp_inproc_plain.log | tail -40
( GL3_TP_DEBUG=1 DBG_ITERS=12 timeout 600 python scripts/debug_tp_vl.py ) > $O/tp_inproc_dbg.log 2>&1; echo "== inproc dbg"; grep -c "gl3 tp dbg" $O/tp_inproc_dbg.log; grep "gl3 tp dbg" $O/tp_inproc_dbg.log | head -20; grep -v "gl3 tp dbg" $O/tp_inproc_dbg.log | tail -30
( GL3_TP_ARENA=cached DBG_ITERS=8 timeout 400 python scripts/debug_tp_vl.py ) > $O/tp_inproc_cached.log 2>&1; echo "== inproc cached arena"; tail -12 $O/tp_inproc_cached.log
( timeout 600 python scripts/debug_tp_procs.py 4 2 25 ) > $O/tp_procs_q4.log 2>&1; echo "== procs q4 x4"; tail -8 $O/tp_procs_q4.log
( timeout 400 python scripts/debug_tp_procs.py 2 8 25 f32act ) > $O/tp_procs_f32.log 2>&1; echo "== procs f32act x2"; tail -8 $O/tp_procs_f32.log
( timeout 300 python scripts/tg_only.py llama-3-8b 32 8 128 ) > $O/tg_graph.log 2>&1; echo "== tg graph"; tail -2 $O/tg_graph.log
( GL3_NO_GRAPH=1 timeout 300 python scripts/tg_only.py llama-3-8b 32 8 128 ) > $O/tg_eager.log 2>&1; echo "== tg eager"; tail -2 $O/tg_eager.log
