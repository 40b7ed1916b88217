#!/bin/bash
# tg128 behind a prefill of d positions (bench.py --depth) for environment variants: scripts/gpu/depth_ab.sh OUTDIR "256,512" "VAR=val" "VAR=val" ...
set -u
O=gpurun_out/$1; D=$2; shift 2; mkdir -p $O
export TMPDIR=/tmp
i=0
for v in "$@"; do
  i=$((i+1))
  ( env $v timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --depth $D > $O/depth_$i.json 2> $O/depth_$i.err; echo "[$v] rc=$?" )
  python3 - <<PY
import json
try:
    d = json.loads(open("$O/depth_$i.json").read().strip().splitlines()[-1])
    print("[$v] tg", d["value"], " ".join("%s %.1f" % (r["test"], r["tok_s"]) for r in d.get("depth_rows", []) if "tok_s" in r), "| pp", d["pp_rows"][0].get("tok_s"))
except Exception as e:
    print("no line", e)
PY
done
