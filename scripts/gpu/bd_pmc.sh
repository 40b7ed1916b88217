#!/bin/bash
# SQ stall breakdown of the small-batch GEMM probe (separate --pmc pass, CSV, bounded)
set -u
O=${1:-gpurun_out/bd_pmc}; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( cd /tmp && timeout 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $R/$O/pmc1 -o p -- $R/scripts/probes/bd_probe 32 > $R/$O/p1.log 2>&1; echo rc=$? )
( cd /tmp && timeout 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $R/$O/pmc2 -o p -- $R/scripts/probes/bd_probe 32 > $R/$O/p2.log 2>&1; echo rc=$? )
ls -R $O | head -30
