"""The A/B forms of the batched-prefill GEMM stay bit-exact: the library reads GL3_PF_GEMM2 / GL3_PF_GEMM2_ALL once per process, so every form
runs the batched-prefill parity tests (tiny shapes to the 8B layer at 512 tokens) in its own pytest subprocess.
   default            r4 kernel (scale products on the matrix pipe) for gate/up, r3 kernel elsewhere   — covered by the normal suite
   GL3_PF_GEMM2_ALL=1 r4 kernel for EVERY shape (qkv / wo / down epilogues: store, residual)
   GL3_PF_GEMM2=1     -B s on the VALU
   GL3_PF_GEMM2=0     r3 kernel everywhere; GL3_PF_FUSED_ATTN=0: the three-kernel prefill attention"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("env", [{"GL3_PF_GEMM2_ALL": "1"}, {"GL3_PF_GEMM2": "1", "GL3_PF_GEMM2_ALL": "1"},
                                 {"GL3_PF_GEMM2": "0", "GL3_PF_FUSED_ATTN": "0"}], ids=["r4-all-shapes", "nbs-on-valu", "r3-kernels"])
def test_prefill_parity_of_a_gemm_form(env):
    e = dict(os.environ, **env)
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_decode.py"), os.path.join(ROOT, "tests", "test_gpu_fullsize.py"),
                          "-m", "gpu", "-x", "-q", "-k", "batched_prefill_is_bit_identical or prefill512", "-p", "no:cacheprovider"],
                         capture_output=True, text=True, timeout=800, env=e, cwd=ROOT)
    tail = out.stdout[-1500:] + out.stderr[-500:]
    assert out.returncode == 0, tail
    assert " passed" in out.stdout and "failed" not in out.stdout, tail
