#!/bin/bash
# Round-3 profile: rocprofv3 kernel statistics of the headline bench (tg + pp), separate --pmc passes (FETCH_SIZE of the decode
# kernels; matrix-pipe and VALU counters + FETCH_SIZE of the batched-prefill GEMMs, Q8_0 and F16), kernel statistics of the B = 32
# step, the VALU-rate probe.  CSV output and a timeout on every rocprofv3 call; --pmc never combined with tracing.
#   bash scripts/gpu/profile_round.sh gpurun_out/prof_rN ; python scripts/summarize_profile.py gpurun_out/prof_rN rNN
set -u
O=${1:-gpurun_out/prof_r3}; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o k -- python $R/bench.py --steps 2 --no-cpu-baseline > $R/$O/bench_traced.json 2> $R/$O/bench_traced.err; echo trace rc=$? )
( cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_fetch -o f -- python $R/bench.py --steps 1 --no-pp --no-cpu-baseline > $R/$O/bench_pmc.json 2> $R/$O/bench_pmc.err; echo pmc rc=$? )
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $R/$O/pmc_pp_q8 -o p -- python $R/scripts/pp_only.py llama-3-8b 4 8 > $R/$O/pp_q8.log 2>&1; echo pp_q8 rc=$? )
( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $R/$O/pmc_pp_q8_fetch -o p -- python $R/scripts/pp_only.py llama-3-8b 4 8 > $R/$O/pp_q8_fetch.log 2>&1; echo pp_q8_fetch rc=$? )
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $R/$O/pmc_pp_f16 -o p -- python $R/scripts/pp_only.py llama-3.2-1b 4 1 > $R/$O/pp_f16.log 2>&1; echo pp_f16 rc=$? )
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_f16 -o k -- python $R/scripts/pp_only.py llama-3.2-1b 16 1 > $R/$O/pp_f16_trace.log 2>&1; echo pp_f16_trace rc=$? )
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_q4 -o k -- python $R/scripts/pp_only.py llama-3-8b 4 2 > $R/$O/pp_q4_trace.log 2>&1; echo pp_q4_trace rc=$? )
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_bd -o bd -- python $R/scripts/bd_only.py qwen3-4b 32 16 > $R/$O/bd.log 2> $R/$O/bd.err; echo bd rc=$? )
( cd $R/scripts/probes && hipcc --offload-arch=gfx950 -O3 valu_rate_probe.hip -o /tmp/valu_rate_probe 2>/dev/null && timeout 60 /tmp/valu_rate_probe > $R/$O/valu_rate_probe.txt 2>&1; echo valu rc=$? )
for spec in "llama-3.2-1b q8_0" "qwen3-4b q8_0"; do
  set -- $spec
  ( timeout 600 python bench.py --steps 3 --warmup 1 --model $1 --wtype $2 --no-cpu-baseline > $O/bench_$1_$2.json 2> $O/bench_$1_$2.err; echo "$1 $2 rc=$?" )
done
# per-dispatch traces are tens of MB (gpurun copies back at most 64 MiB): keep the statistics and counter tables, drop the traces
find $O -name "*kernel_trace.csv" -delete
du -sh $O; cat $O/pp_q8.log $O/pp_f16.log | grep pp512; tail -2 $O/valu_rate_probe.txt
