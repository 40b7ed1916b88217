#!/bin/bash
# A/B of two builds of the library (GL3_LIB): 8B Q8_0 bench line, 8B Q4_0, 1B F16, Qwen3-4B B=32.  $2 = library file under gpullama3.java_amd/
set -u
O=gpurun_out/${1:-r3ab}; mkdir -p $O
export TMPDIR=/tmp
for lib in libgpullama_hip.so ${2:-libgpullama_hip_nopk.so}; do
  export GL3_LIB=$PWD/gpullama3.java_amd/$lib
  tag=${lib%.so}; tag=${tag#libgpullama_}
  echo "=== $lib"
  for spec in "llama-3-8b q8_0" "llama-3-8b q4_0" "llama-3.2-1b f16" ${AB_EXTRA:-}; do
    set -- $spec
    ( timeout 600 python bench.py --steps 2 --warmup 1 --model $1 --wtype $2 --no-cpu-baseline > $O/bench_${tag}_$1_$2.json 2> $O/bench_${tag}_$1_$2.err; echo "$1 $2 rc=$?" )
    python - <<PY
import json
try:
    d=json.load(open("$O/bench_${tag}_$1_$2.json")); print("$1 $2: tg", d["value"], "pp", [(r.get("batch"), r.get("tok_s", r.get("error"))) for r in d["pp_rows"]], {k:v["avg_us"] for k,v in d.get("roofline_pp",{}).get("gemms",{}).items()})
except Exception as e: print("no json", e)
PY
  done
  ( timeout 600 python bench.py --model qwen3-4b --decode-batch 32 --steps 2 --warmup 1 > $O/bench_${tag}_bd32.json 2> $O/bench_${tag}_bd32.err; echo "bd32 rc=$?" )
  python - <<PY
import json
d=json.load(open("$O/bench_${tag}_bd32.json")); print("bd32", d["value"], d["ms_per_batched_step"])
PY
done
