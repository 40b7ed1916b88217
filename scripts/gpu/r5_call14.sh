#!/bin/bash
# round 5, call 14: pipelined MFMA-product GEMM: parity (f32-activation batched tests), pp512 on 8 layers, per-kernel times
set -u
O=gpurun_out/r5_call14; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 1500 python -m pytest tests/test_gpu_decode.py tests/test_gpu_fullsize.py tests/test_gpu_tp.py -m gpu -x -q --timeout 900 -k "f32_activation or q4 or Q4 or vl or 8b_layer or prefill or batched" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -12 ) > $O/pytest.log 2>&1; echo "== pytest"; tail -4 $O/pytest.log
timeout 300 python scripts/pp_only.py llama-3-8b 8 2 2>&1 | tail -1
echo "== q8 f32act (type 8 + flag?)"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o k -- python $R/scripts/pp_only.py llama-3-8b 2 2 > $R/$O/trace.log 2>&1 )
python - <<'PY'
import csv, glob, collections
fs = glob.glob("gpurun_out/r5_call14/trace/**/*kernel_trace.csv", recursive=True)
if fs:
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "gl3::" in r["Kernel_Name"] or "pf_" in r["Kernel_Name"]:
            d[(r["Kernel_Name"][:70], r.get("Grid_Size_X", r.get("Grid_Size")))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:14]: print(k, len(v), "avg us %.1f" % (sum(v) / len(v)))
PY
