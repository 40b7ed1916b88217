#!/bin/bash
# Batched-prefill GEMM classes (scripts/gemm_ab.py: one HIP event pair per class + pp512 wall time) over builds of the library.
#   scripts/gpu/gemm_libs_ab.sh OUTDIR "lib[:VAR=val,VAR=val]" ...      lib = file under gpullama3.java_amd/ (make variant V=...)
# MODEL / NLAYERS / NTOK choose the shape (default llama-3-8b 4 512); REPEAT = round-robin passes over the list (default 1).  Process-to-process
# spread on one box is up to 20 % (clock state), so compare the per-variant MINIMUM over several passes, printed at the end.  Every command is bounded.
set -u
O=gpurun_out/$1; shift; mkdir -p $O
export TMPDIR=/tmp
for rep in $(seq 1 ${REPEAT:-1}); do
for spec in "$@"; do
  lib=${spec%%:*}; envs=""
  [ "$spec" != "$lib" ] && envs=$(echo "${spec#*:}" | tr ',' ' ')
  ( env GL3_LIB=$PWD/gpullama3.java_amd/$lib $envs timeout 300 python scripts/gemm_ab.py ${MODEL:-llama-3-8b} ${NLAYERS:-4} ${NTOK:-512} 2>&1 | tail -${TAILN:-1} | sed "s|^|$lib |" ) >> $O/ab.log 2>&1
done
done
python3 - $O/ab.log <<'PY'
import re, sys, collections
rows = collections.defaultdict(list)
for l in open(sys.argv[1]):
    m = re.match(r"(\S+) \[(.*?)\] (.*) \| pp\d+ \d+ layers: [\d.]+ ms = ([\d.]+) us/layer", l)
    if not m: continue
    tag = m.group(1) + " " + " ".join(x for x in m.group(2).split() if not x.startswith("GL3_LIB="))
    vals = [float(x) for x in re.findall(r"([\d.]+) us \(", m.group(3))] + [float(m.group(4))]
    rows[tag].append(vals)
print("min over passes (qkv, wo, gate/up, down, pp us/layer) | median")
for tag, v in rows.items():
    mins = [min(c) for c in zip(*v)]; meds = [sorted(c)[len(c) // 2] for c in zip(*v)]
    print("%-70s n=%d  min %s | med %s" % (tag, len(v), " ".join("%6.1f" % x for x in mins), " ".join("%6.1f" % x for x in meds)))
PY
