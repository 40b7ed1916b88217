#!/bin/bash
# round 5, call 19: bank-conflict-free product layout in attn_pv_kernel: parity at depth, depth rows, LDS counters
set -u
O=gpurun_out/r5_call19; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_gpu_decode.py -m gpu -x -q --timeout 600 -k "context or depth or handover or long or 21" 2>&1 | tail -3 ) > $O/pytest.log 2>&1; echo "== pytest"; tail -2 $O/pytest.log
( timeout 600 python bench.py --steps 3 --warmup 1 --depth 1024,4096,16384 --no-cpu-baseline --no-pp 2> $O/bench_depth.err | tail -1 ) > $O/bench_depth.json; python - <<PY
import json
d = json.load(open("$O/bench_depth.json")); print("tg128", d["value"])
for r in d.get("depth_rows", []): print(r["test"], r.get("tok_s"), r.get("attention_us_per_layer"))
PY
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/$O/p1 -o p -- python $R/scripts/depth_only.py llama-3-8b 8 16384 18 > $R/$O/p1.log 2>&1 )
python scripts/pmc_table.py $O/p1 attn_pv > $O/pmc_pv.csv; cat $O/pmc_pv.csv | cut -c1-200
find $O/p1 -name "*.csv" -size +2M -delete
