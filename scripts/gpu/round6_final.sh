#!/bin/bash
# Round-6 closing session: the GPU test suite file by file, the headline bench line, rocprofv3 kernel statistics + separate --pmc passes
# (never combined with tracing), and >= 5-step bench lines WITH a cpu_baseline leg for the other configurations.  Every command bounded.
#   bash scripts/gpu/round6_final.sh gpurun_out/r6_final [tests|bench|prof|lines ...]    (no stage list = all)
set -u
O=${1:-gpurun_out/r6_final}; shift || true
STAGES=${*:-tests bench prof lines depth}
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for st in $STAGES; do case $st in
tests)
  for f in gpu_tp gpu_decode gpu_fullsize gpu_kquant gpu_sampling gpu_seqsum gpu_run_host gpu_moe gpu_gemm_forms reference_golden; do
    ( timeout 600 python -m pytest tests/test_$f.py -m gpu -x -q --timeout 240 2>&1 | grep -E "passed|failed|skipped|^FAILED|rror" | tail -4 ) > $O/pytest_$f.log 2>&1
    echo "== $f: $(tail -1 $O/pytest_$f.log)"
  done ;;
bench)
  ( timeout 600 python bench.py > $O/bench_8b.json 2> $O/bench_8b.err; echo "bench rc=$?" )
  python - <<PY
import json
d = json.load(open("$O/bench_8b.json"))
print("8B: tg", d["value"], "pp", [(r["batch"], r.get("tok_s")) for r in d["pp_rows"]], "roofline", d["roofline"]["frac"], d["roofline"].get("frac_of_peak_measured"),
      "pp gemms", {k: v["avg_us"] for k, v in d["roofline_pp"]["gemms"].items()}, "cpu", d["cpu_baseline"]["value"])
PY
  ;;
prof)
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o k -- python $R/bench.py --steps 2 --no-cpu-baseline > $R/$O/bench_traced.json 2> $R/$O/bench_traced.err; echo trace rc=$? )
  ( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_fetch -o f -- python $R/bench.py --steps 1 --no-pp --no-cpu-baseline > $R/$O/bench_pmc.json 2> $R/$O/bench_pmc.err; echo pmc rc=$? )
  ( cd /tmp && timeout 240 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $R/$O/pmc_pp_q8 -o p -- python $R/scripts/pp_only.py llama-3-8b 4 8 > $R/$O/pp_q8.log 2>&1; echo pp_q8 rc=$? )
  ( cd /tmp && timeout 240 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $R/$O/pmc_pp_q8_fetch -o p -- python $R/scripts/pp_only.py llama-3-8b 4 8 > $R/$O/pp_q8_fetch.log 2>&1; echo pp_q8_fetch rc=$? )
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_bd -o bd -- python $R/scripts/bd_only.py qwen3-4b 32 16 > $R/$O/bd.log 2> $R/$O/bd.err; echo bd rc=$? )
  find $O -name "*kernel_trace.csv" -delete
  cat $O/pp_q8.log | grep pp512 ;;
lines)
  for spec in "qwen3-4b q8_0" "llama-3.2-1b q8_0" "llama-3-8b q4_0" "llama-3.2-1b f16" "llama-3-8b q8_0_f32act"; do
    set -- $spec
    ( timeout 500 python bench.py --steps 5 --warmup 1 --model $1 --wtype $2 --cpu-seconds 10 > $O/bench_$1_$2.json 2> $O/bench_$1_$2.err; echo "$1 $2 rc=$?" )
    python - <<PY
import json
try:
    d = json.load(open("$O/bench_$1_$2.json")); print("$1 $2: tg", d["value"], "pp", [(r.get("batch"), r.get("tok_s", r.get("error"))) for r in d["pp_rows"]], "roofline", d["roofline"]["frac"], "cpu", d["cpu_baseline"]["value"])
except Exception as e: print("no json", e)
PY
  done
  ( timeout 500 python bench.py --model qwen3-4b --decode-batch 32 --steps 5 --warmup 1 --cpu-seconds 10 > $O/bench_qwen3-4b_bd32.json 2> $O/bench_qwen3-4b_bd32.err; echo "bd32 rc=$?" )
  python - <<PY
import json
try:
    d = json.load(open("$O/bench_qwen3-4b_bd32.json")); print("bd32", d["value"], d["ms_per_batched_step"], d["roofline"]["frac"], d["cpu_baseline"] and d["cpu_baseline"]["value"])
except Exception as e: print("no json", e)
PY
  ;;
depth)
  # r6: mid-regime rows included
  ( timeout 600 python bench.py --steps 3 --warmup 1 --depth 256,512,1024,4096,16384 --no-pp > $O/bench_8b_depth.json 2> $O/bench_8b_depth.err; echo "depth rc=$?" )
  python - <<PY
import json
try:
    d = json.load(open("$O/bench_8b_depth.json")); print("depth:", [(r["test"], r.get("tok_s"), r.get("attention_us_per_layer")) for r in d["depth_rows"]])
except Exception as e: print("no json", e)
PY
  DEPTHS="4096 16384" OUT=$(basename $O)/prof_depth bash scripts/gpu/prof_depth.sh 2>&1 | grep "gl3::attn\|== depth"
  ( GL3_CTX=8192 timeout 900 python scripts/native_bench_8b.py llama-3-8b -p 512 -n 128 -pg 512,128 -d 0,4096 -b 512 -r 3 -o jsonl > $O/native_bench_8b.log 2>&1; echo "native rc=$?"; tail -8 $O/native_bench_8b.log )
  ;;
esac; done
du -sh $O
