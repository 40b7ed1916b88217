#!/bin/bash
set -u
O=gpurun_out/r2m; mkdir -p $O
export TMPDIR=/tmp
( GL3_LIB=$GRAFT_REPO_ROOT/gpullama3.java_amd/libgpullama_hip_scalar_epi.so timeout 600 python -m pytest tests/test_gpu_decode.py -m gpu -x -q -k "batched_prefill" 2>&1 | tail -3 ) > $O/pytest_b.log 2>&1
tail -2 $O/pytest_b.log
for v in A B; do
  if [ $v = B ]; then export GL3_LIB=$GRAFT_REPO_ROOT/gpullama3.java_amd/libgpullama_hip_scalar_epi.so; fi
  ( timeout 300 python bench.py --steps 3 --no-cpu-baseline --n-gen 16 > $O/bench_pp_$v.json 2> $O/bench_pp_$v.err )
  python - "$O/bench_pp_$v.json" "$v" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], [ (r["batch"], r.get("tok_s")) for r in d["pp_rows"]], {k:v["avg_us"] for k,v in d["roofline_pp"]["gemms"].items()})
except Exception as e: print("ERR", sys.argv[2], e)
PY
done
