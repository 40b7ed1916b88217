#!/bin/bash
# Batched-prefill GEMM classes (scripts/gemm_ab.py: one HIP event pair per class + pp512 wall time) over builds of the library.
#   scripts/gpu/gemm_libs_ab.sh OUTDIR "lib[:VAR=val,VAR=val]" ...      lib = file under gpullama3.java_amd/ (make variant V=...)
# MODEL / NLAYERS / NTOK choose the shape (default llama-3-8b 4 512).  Every command is bounded.
set -u
O=gpurun_out/$1; shift; mkdir -p $O
export TMPDIR=/tmp
for spec in "$@"; do
  lib=${spec%%:*}; envs=""
  [ "$spec" != "$lib" ] && envs=$(echo "${spec#*:}" | tr ',' ' ')
  ( env GL3_LIB=$PWD/gpullama3.java_amd/$lib $envs timeout 300 python scripts/gemm_ab.py ${MODEL:-llama-3-8b} ${NLAYERS:-4} ${NTOK:-512} 2>&1 | tail -${TAILN:-1} | sed "s|^|$lib |" ) >> $O/ab.log 2>&1
done
cat $O/ab.log
