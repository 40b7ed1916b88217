#!/bin/bash
# long-context windows + roctx ranges: parity tests, then a marker trace of two eager decode steps (GL3_ROCTX=1)
set -u
O=gpurun_out/${1:-ctxroctx}; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 1500 python -m pytest tests/test_gpu_decode.py -m gpu -x -q -k "long_context or beyond_20k or handover or live or sequential_prefill" 2>&1 | tail -6 ) > $O/pytest.log 2>&1
cat $O/pytest.log
( cd /tmp && GL3_ROCTX=1 timeout 300 rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d $R/$O/roctx -o m -- python $R/scripts/tg_only.py llama-3-8b 2 8 4 > $R/$O/roctx.log 2>&1; echo "roctx rc=$?"; tail -2 $R/$O/roctx.log )
find $O -name "*kernel_trace.csv" -delete
ls $O/roctx | head; for f in $O/roctx/*marker*stats*.csv $O/roctx/*marker_api_stats.csv; do [ -f "$f" ] && { echo "== $f"; head -20 "$f"; }; done
