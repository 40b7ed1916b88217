#!/bin/bash
# bench.py as the driver launches it for N = 2, both ranks on the one GPU of the box (GL3_BENCH_SHARE_GPU): exercises the
# gloo control plane, the IPC handle exchange, the transport self-test and the peer-write gathers between two processes.
set -u
O=${1:-gpurun_out/tp2}; mkdir -p $O
export TMPDIR=/tmp GL3_BENCH_SHARE_GPU=1 GPU_MAX_HW_QUEUES=8
( timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 2 --warmup 1 --model llama-3.2-1b --no-cpu-baseline > $O/bench_tp2.json 2> $O/bench_tp2.err; echo rc=$? )
tail -c 1500 $O/bench_tp2.json; tail -5 $O/bench_tp2.err
