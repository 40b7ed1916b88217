#!/bin/bash
set -u
export TMPDIR=/tmp
for m in 0 1; do echo "== GL3_VQM_OCC4=$m"; GL3_VQM_OCC4=$m timeout 300 python scripts/pp_only.py llama-3-8b 8 2 2>&1 | tail -1; done
