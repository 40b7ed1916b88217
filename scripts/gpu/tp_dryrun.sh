#!/bin/bash
# bench.py exactly as the driver launches it for N = 2 / 4 / 8 — but with every rank on the ONE GPU of the box (GL3_BENCH_SHARE_GPU): the gloo control
# plane, the IPC handle exchange, the transport self-test, the folded peer-write hand-overs and the JSON line of the multi-rank path run before the
# driver's SCALE run does.  The tok/s of these lines mean nothing (N processes time-share one GPU); ranks / transport / fold_mode / parity do.
#   scripts/gpu/tp_dryrun.sh OUTDIR [model] [wtype] ["N N N"]
set -u
O=${1:-gpurun_out/tp_dryrun}; M=${2:-llama-3-8b}; W=${3:-q4_0}; NS=${4:-"2 4 8"}; mkdir -p $O
export TMPDIR=/tmp GL3_BENCH_SHARE_GPU=1 GPU_MAX_HW_QUEUES=32
port=29560
for n in $NS; do
  port=$((port + 1))
  ( timeout ${TP_TIMEOUT:-900} python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port bench.py --gpus $n --steps 2 --warmup 1 \
      --model $M --wtype $W --no-cpu-baseline > $O/tp_dryrun_${M}_${W}_n$n.json 2> $O/tp_dryrun_${M}_${W}_n$n.err; echo "n=$n rc=$?" )
  python3 - <<PY
import json
try:
    d = json.loads(open("$O/tp_dryrun_${M}_${W}_n$n.json").read().strip().splitlines()[-1])
    print("n=$n", d["value"], "tok/s (shared GPU)", d["config"]["parallelism"], "ranks", d["config"]["ranks"], "transport", d["config"]["transport"], "fold", d["config"]["fold_mode"])
except Exception as e:
    print("n=$n no line:", e); print(open("$O/tp_dryrun_${M}_${W}_n$n.err").read()[-1500:])
PY
done
