#!/bin/bash
# r4: per-phase cycle stamps of diagnostic builds of pf_gemm2_kernel ($2.. = variant names: libgpullama_hip_<name>.so)
set -u
O=$1; shift; mkdir -p $O
export TMPDIR=/tmp
for v in "$@"; do
  export GL3_LIB=$PWD/gpullama3.java_amd/libgpullama_hip_$v.so
  ( timeout 300 python scripts/gemm_ab.py llama-3-8b 1 2>&1 | grep -E "g2 EPI|pp512" ) > $O/timing_$v.log 2>&1
  python - <<PY
import re, collections
acc = collections.defaultdict(list)
for ln in open("$O/timing_$v.log"):
    m = re.search(r"g2 EPI (\d) NW (\d) J (\d+) wave (\d) stages (\d+): issue (\d+) compute (\d+) store\+wait (\d+) barrier (\d+) total (\d+)", ln)
    if m:
        acc[(m.group(1), m.group(2), m.group(5))].append([int(x) for x in m.groups()[5:]])
    elif "pp512" in ln: print("$v", ln.strip()[:200])
for k, v in sorted(acc.items()):
    n = len(v)
    print("$v EPI", k[0], "NW", k[1], "stages", k[2], "samples", n, "per stage: issue %.0f compute %.0f store+wait %.0f barrier %.0f | total %.0f" % tuple([sum(x[i] for x in v) / n / int(k[2]) for i in range(4)] + [sum(x[4] for x in v) / n]))
PY
  rm -f $O/timing_$v.log
done
