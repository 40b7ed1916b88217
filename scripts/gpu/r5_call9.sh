#!/bin/bash
# round 5, call 9: folded tensor-parallel gathers (GL3_TP_FOLD = 0 gather kernels / 1 push + wait launch / 2 wait inside consumers)
set -u
O=gpurun_out/r5_call9; mkdir -p $O
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_gpu_tp.py -m gpu -x -q --timeout 900 2>&1 | tail -8 ) > $O/pytest_fold1.log 2>&1; echo "== pytest tp (fold 1)"; tail -4 $O/pytest_fold1.log
( GL3_TP_FOLD=2 GL3_TP_SPIN_LIMIT=2000000 timeout 900 python -m pytest tests/test_gpu_tp.py -m gpu -x -q --timeout 600 -k "row_split_ranks or peer_write" 2>&1 | tail -8 ) > $O/pytest_fold2.log 2>&1; echo "== pytest tp (fold 2)"; tail -4 $O/pytest_fold2.log
for mode in 0 1; do
  export GL3_TP_FOLD=$mode
  bash scripts/gpu/tp2_bench.sh $O/tp2_fold$mode > $O/tp2_fold$mode.log 2>&1
  echo "== tp2 bench GL3_TP_FOLD=$mode"; python - <<PY
import json
try:
    d = json.loads(open("$O/tp2_fold$mode/bench_tp2.json").read().strip().splitlines()[-1])
    print("tok/s", d["value"], "ms/step", d["ms_per_step"], d["config"].get("parallelism"))
except Exception as e:
    print("no json:", e); print(open("$O/tp2_fold$mode.log").read()[-1500:])
PY
done
unset GL3_TP_FOLD
( timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pp 2> $O/bench8b.err | tail -1 ) > $O/bench8b.json; python -c "
import json; d=json.load(open('$O/bench8b.json')); print('8B tg', d['value'], d.get('roofline'))"
