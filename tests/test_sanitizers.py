"""the host-side code under AddressSanitizer + UndefinedBehaviorSanitizer (scripts/sanitize.sh) — the oracle's KAT /
cross-check / sampling tests on an instrumented build of oracle/gl3_oracle.c, and the native GGUF reader walking valid, truncated and
corrupted files (rejected with an error, never a crash or a sanitizer report).  CPU only, ~20 s."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_and_gguf_reader_are_clean_under_asan_ubsan(tmp_path):
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan) or not os.path.exists("/opt/rocm/lib/llvm/bin/clang++") or shutil.which("make") is None:
        pytest.skip("no sanitizer runtime / host clang++ in this image")
    out = subprocess.run(["bash", os.path.join(ROOT, "scripts", "sanitize.sh"), str(tmp_path)], capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert " passed" in out.stdout and "failed" not in out.stdout
    assert "0 crashes" in out.stdout and "rejected cleanly" in out.stdout
