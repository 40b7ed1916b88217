#!/bin/bash
# round 5, call 11: folded gathers after the fence change: parity (modes 1, 2 with small grids), two-process step time per mode
set -u
O=gpurun_out/r5_call11; mkdir -p $O
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_gpu_tp.py -m gpu -x -q --timeout 900 2>&1 | tail -4 ) > $O/pytest_fold1.log 2>&1; echo "== pytest tp (fold 1)"; tail -2 $O/pytest_fold1.log
( GL3_TP_FOLD=2 GL3_WGS=16 GL3_TP_SPIN_LIMIT=2000000 timeout 900 python -m pytest tests/test_gpu_tp.py -m gpu -x -q --timeout 600 -k "row_split_ranks or peer_write" 2>&1 | tail -4 ) > $O/pytest_fold2.log 2>&1; echo "== pytest tp (fold 2, GL3_WGS=16)"; tail -2 $O/pytest_fold2.log
for mode in 0 1 7; do
  if [ $mode = 7 ]; then export GL3_TP_FOLD=2 GL3_TP_FOLD_MASK=7; else export GL3_TP_FOLD=$mode; fi
  bash scripts/gpu/tp2_bench.sh $O/tp2_fold$mode > $O/tp2_fold$mode.log 2>&1
  echo "== tp2 bench fold $mode"; python - <<PY
import json
try:
    d = json.loads(open("$O/tp2_fold$mode/bench_tp2.json").read().strip().splitlines()[-1])
    print("tok/s", d["value"], "ms/step", d["ms_per_step"])
except Exception as e:
    print("no json:", e); print(open("$O/tp2_fold$mode.log").read()[-1500:])
PY
done
