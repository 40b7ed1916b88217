#!/bin/bash
# SQ / LDS / TCP counters of the batched-prefill GEMMs (separate --pmc passes, no tracing), variant = $2 ("VAR=val ...")
#   CMD="python scripts/pp_only.py llama-3-8b 2 8" FILTER="pf_scores pf_pv" PASSES="1 2" bash scripts/gpu/gemm_pmc.sh out "PP_DEPTH=4096"   (any command / kernels)
set -u
O=$1; V=${2:-GL3_NOOP=1}; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_MISC"
P3="SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES GRBM_GUI_ACTIVE"
P4="FETCH_SIZE TCP_TCC_READ_REQ_sum"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  [ -n "${PASSES:-}" ] && case " $PASSES " in *" $((i+1)) "*) ;; *) i=$((i+1)); continue;; esac
  i=$((i+1))
  ( cd /tmp && env $V timeout 300 rocprofv3 --pmc $P --output-format csv -d $R/$O/p$i -o p -- ${CMD:-python $R/scripts/gemm_ab.py llama-3-8b 2} > $R/$O/p$i.log 2>&1; echo "pass $i rc=$?" )
  python scripts/pmc_table.py $O/p$i ${FILTER:-gemm} > $O/pmc_p$i.csv 2>> $O/p$i.log
  find $O/p$i -name "*.csv" -size +2M -delete
done
tail -2 $O/p1.log; cat $O/pmc_p*.csv
