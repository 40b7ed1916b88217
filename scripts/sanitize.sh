#!/bin/bash
# Sanitizer pass over the host-side code: AddressSanitizer + UBSan builds of
#   oracle/gl3_oracle.c         -> the oracle's KAT / cross-check / sampling tests run on it (libasan preloaded into python)
#   csrc/gl3_gguf.cpp           -> tools/gguf_check walks valid GGUF files and truncated / corrupted copies (must be rejected, never crash)
# No GPU needed.  Exit code 0 = clean.   bash scripts/sanitize.sh [workdir]
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=${1:-/tmp/gl3_sanitize}; mkdir -p $W
cd $ROOT
make -s -C oracle asan
make -s -C gpullama3.java_amd/csrc asan
ASAN=$(gcc -print-file-name=libasan.so)
echo "== oracle under ASan + UBSan"
LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 GL3_ORACLE_LIB=$ROOT/oracle/_asan/libgl3_oracle.so \
  timeout 900 python -m pytest tests/test_oracle_kat.py tests/test_oracle_cross.py tests/test_sampling.py -q -x -p no:cacheprovider 2>&1 | tail -3
echo "== GGUF reader under ASan + UBSan"
python - "$W" <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge
pkg = ge.load_package()
w = sys.argv[1]
for cfg, wt in (("tiny-llama", 8), ("tiny-qwen3", 8), ("tiny-llama-tied", 2), ("tiny-phi3", 1)):
    pkg.synth.make_numpy(pkg.synth.CONFIGS[cfg], wtype=wt, seed=3).write_gguf(os.path.join(w, "%s_%d.gguf" % (cfg, wt)))
b = open(os.path.join(w, "tiny-llama_8.gguf"), "rb").read()
for i, cut in enumerate((0, 3, 11, 24, 200, 4000, len(b) // 2, len(b) - 1)):       # truncations: header, key/value table, tensor table, data
    open(os.path.join(w, "trunc_%d.gguf" % i), "wb").write(b[:cut])
import random
rnd = random.Random(5)
for i in range(24):                                                                 # byte flips inside the first 6 KB (header + tables)
    c = bytearray(b)
    for _ in range(4):
        c[rnd.randrange(0, min(len(c), 6000))] = rnd.randrange(256)
    open(os.path.join(w, "flip_%d.gguf" % i), "wb").write(bytes(c))
PY
ok=0; rej=0
for f in $W/*.gguf; do
  set +e
  ASAN_OPTIONS=detect_leaks=1:abort_on_error=0 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 timeout 60 build_asan/gguf_check $f > $W/out.txt 2>&1
  rc=$?
  set -e
  if [ $rc -eq 0 ]; then ok=$((ok+1)); elif [ $rc -eq 3 ]; then rej=$((rej+1)); else echo "SANITIZER / CRASH on $f (rc $rc)"; tail -20 $W/out.txt; exit 1; fi
done
echo "gguf_check: $ok files parsed, $rej rejected cleanly, 0 crashes"
