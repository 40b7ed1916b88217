#!/bin/bash
# r4: per-phase cycle stamps of pf_gemm2_kernel (diagnostic build libgpullama_hip_timing.so): one gate/up sweep on 2 layers
set -u
O=$1; mkdir -p $O
export TMPDIR=/tmp
export GL3_LIB=$PWD/gpullama3.java_amd/libgpullama_hip_timing.so
( timeout 300 python scripts/gemm_ab.py llama-3-8b 1 2>&1 | grep -E "g2 EPI 2|pp512" | sort | uniq -c | sort -k1,1nr | head -60 ) > $O/timing.log 2>&1
python - <<PY
import re, collections
acc = collections.defaultdict(list)
for ln in open("$O/timing.log"):
    m = re.search(r"g2 EPI (\d) NW (\d) J (\d+) wave (\d) stages (\d+): issue (\d+) compute (\d+) store\+wait (\d+) barrier (\d+) total (\d+)", ln)
    if m:
        acc[(m.group(1), m.group(5))].append([int(x) for x in m.groups()[5:]])
for k, v in acc.items():
    n = len(v)
    print("EPI", k[0], "stages", k[1], "samples", n, "mean cycles: issue %.0f compute %.0f store+wait %.0f barrier %.0f total %.0f" % tuple(sum(x[i] for x in v) / n for i in range(5)))
PY
tail -3 $O/timing.log
