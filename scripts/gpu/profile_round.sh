#!/bin/bash
# Round profile: rocprofv3 kernel statistics (tg + pp of the headline bench) and, in a SEPARATE pass, the FETCH_SIZE counter of
# the decode kernels.  CSV output and a timeout on every rocprofv3 call (the default rocpd output once hung after finalisation).
#   bash scripts/gpu/profile_round.sh gpurun_out/prof_r2 ; python scripts/summarize_profile.py gpurun_out/prof_r2 r02
set -u
O=${1:-gpurun_out/prof_round}; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o k -- python $R/bench.py --steps 2 --no-cpu-baseline > $R/$O/bench_traced.json 2> $R/$O/bench_traced.err; echo trace rc=$? )
( cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_fetch -o f -- python $R/bench.py --steps 1 --no-pp --no-cpu-baseline > $R/$O/bench_pmc.json 2> $R/$O/bench_pmc.err; echo pmc rc=$? )
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_bd -o bd -- python $R/scripts/bd_only.py qwen3-4b 32 16 > $R/$O/bd.log 2> $R/$O/bd.err; echo bd rc=$? )
# per-dispatch traces are tens of MB (gpurun copies back at most 64 MiB): keep the statistics, drop the traces
find $O -name "*kernel_trace.csv" -delete
find $O -name "*.csv" | head -20; du -sh $O
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); grep -v "at::native\|rocclr" $f | head -14 | cut -c1-160
