#!/bin/bash
set -u
O=${1:-gpurun_out/f32act}; mkdir -p $O
( timeout 600 python bench.py --wtype q8_0_f32act --cpu-seconds 8 > $O/bench_8b_q8_0_f32act.json 2> $O/bench.err ); tail -3 $O/bench.err
python - $O/bench_8b_q8_0_f32act.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], d["roofline"]["frac"], d.get("cpu_baseline"))
PY
