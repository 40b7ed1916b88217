"""The forms of the batched-prefill GEMM stay bit-exact.  The library reads its switches once per process, so every form runs the > 64-token prefill
parity tests (ragged small shapes to the 8B / 1B layers at 512 tokens) in its own pytest subprocess.
   default                     r6: pf_gemm3_kernel for qkv / wo / down, the tall one-round tiling (pf_gemm3t_kernel) or the 128 x 128 tiling for gate + up — normal suite
   GL3_PF_GEMM3_TALL=-1        gate + up on the 128 x 128 tiling everywhere;  =4..7 the tall tiling with that many row fragments on EVERY shape
   GL3_PF_GEMM3_TALL_KB=1      one block per K stage of the tall tiling (default: two)
   GL3_PF_GEMM3_SHAPE=1|2|3    128 x 128 / 96 x 128 / 64 x 128 workgroup tiles for every non-SwiGLU projection
   GL3_PF_GEMM3=0              the r5 kernels (row-layout activations): r4 kernel for gate + up, r3 kernel elsewhere; with GL3_PF_GEMM2=0 the r3 kernel
                               everywhere, GL3_PF_FUSED_ATTN=0 the three-kernel prefill attention"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("env", [{"GL3_PF_GEMM3": "0"}, {"GL3_PF_GEMM3_TALL": "-1"}, {"GL3_PF_GEMM3_TALL": "4"}, {"GL3_PF_GEMM3_TALL": "5", "GL3_PF_GEMM3_TALL_KB": "1"},
                                 {"GL3_PF_GEMM3_TALL": "7", "GL3_PF_GEMM3_SHAPE": "1"}, {"GL3_PF_GEMM3_TALL": "6", "GL3_PF_GEMM3_SHAPE": "2"}],
                         ids=["r5-kernels", "g3-128x128", "tall4", "tall5-one-block-stages", "tall7-shape1", "tall6-shape2"])
def test_prefill_parity_of_a_gemm_form(env):
    e = dict(os.environ, **env)
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_decode.py"), os.path.join(ROOT, "tests", "test_gpu_fullsize.py"),
                          "-m", "gpu", "-x", "-q", "-k", "chunks_above_64 or prefill512", "-p", "no:cacheprovider"],
                         capture_output=True, text=True, timeout=800, env=e, cwd=ROOT)
    tail = out.stdout[-1500:] + out.stderr[-500:]
    assert out.returncode == 0, tail
    assert " passed" in out.stdout and "failed" not in out.stdout, tail


@pytest.mark.parametrize("env,select", [({"GL3_BDK_P": "3"}, "static_batched or batched_prefill_is_bit or b32"),
                                        ({"GL3_BDK_P": "2", "GL3_BDK_DA": "8", "GL3_BDK_GU": "0"}, "static_batched or b32")],
                         ids=["three-producers", "two-producers-ring-of-8"])
def test_k_split_small_batch_gemm_stays_bit_exact(env, select):
    """GL3_BDK=1: the static-batched decode / small-chunk GEMM with K split over producer wavefronts and an ordered chain wavefront (gl3_bdk_gemm.h,
    off by default: measured slower) against the same parity tests as the default one-wavefront-per-tile kernel; the second form runs the ring of
    8 tiles per producer (single-matrix classes) beside the default gate + up kernel."""
    e = dict(os.environ, GL3_BDK="1", **env)
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_decode.py"), os.path.join(ROOT, "tests", "test_gpu_fullsize.py"),
                          "-m", "gpu", "-x", "-q", "-k", select, "-p", "no:cacheprovider"],
                         capture_output=True, text=True, timeout=800, env=e, cwd=ROOT)
    tail = out.stdout[-1500:] + out.stderr[-500:]
    assert out.returncode == 0, tail
    assert " passed" in out.stdout and "failed" not in out.stdout, tail


@pytest.mark.parametrize("env,select", [
    ({"GL3_PF_FUSED_ATTN": "0"}, "chunks_above_64 or prefill512 or long_context_prefill or behind_1000 or batched_prefill_is_bit"),
    ({"GL3_PF_FUSED_ATTN": "0", "GL3_PF_SOFTMAX_ROWS": "0"}, "chunks_above_64 or long_context_prefill or behind_1000 or batched_prefill_is_bit"),
    ({"GL3_PF_FUSED_ATTN": "0", "GL3_PF_SCORES_MFMA": "0", "GL3_PF_PV_MFMA": "0"}, "chunks_above_64 or long_context_prefill or behind_1000"),
    ({"GL3_PF_FUSED_ATTN": "0", "GL3_PF_SCORES_MFMA": "0", "GL3_PF_SCORES_PK": "0", "GL3_PF_PV_MFMA": "0", "GL3_PF_PV_RING": "0"}, "chunks_above_64 or long_context_prefill"),
    ({"GL3_PF_FUSED_MFMA": "0"}, "chunks_above_64 or llama3_8b_shaped_layer_prefill512"),
    ({"GL3_PF_FUSED_V1": "1"}, "chunks_above_64 or long_context_prefill")],
    ids=["three-kernels", "three-kernels-row-per-wavefront", "three-kernels-valu-packed", "three-kernels-valu-scalar", "one-launch-valu-packed", "one-launch-r4"])
def test_prefill_attention_forms(env, select):
    """The forms of the prefill attention stay bit-exact, each in its own process (switches are read once):
       default                      one launch with MFMA products (pf_attn_fused3_kernel) while a tile's score rows fit LDS, else pf_scores_mfma_kernel ->
                                    pf_softmax_rows_kernel -> pf_pv_mfma_kernel (kvMul 4, head size 128 / 64; other shapes: the VALU kernels)
       GL3_PF_FUSED_ATTN=0          the three kernels from position 0
       GL3_PF_SOFTMAX_ROWS=0        the r1 softmax kernel (one row per wavefront, normalises in place) and the r1 V-sum kernel
       GL3_PF_SCORES_MFMA=0 / GL3_PF_PV_MFMA=0     packed-f32 VALU products (pf_scores_pk_kernel, pf_pv_ring_kernel); + _PK=0 / _RING=0: the scalar r1 kernels
       GL3_PF_FUSED_MFMA=0 / GL3_PF_FUSED_V1=1     the one-launch kernel with packed-f32 VALU products (fused2) / the r4 kernel
    Ragged chunks, chunks at non-zero positions, the 8B / 1B layers at 512 tokens and behind 1000 positions."""
    e = dict(os.environ, **env)
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_decode.py"), os.path.join(ROOT, "tests", "test_gpu_fullsize.py"),
                          "-m", "gpu", "-x", "-q", "-k", select, "-p", "no:cacheprovider"],
                         capture_output=True, text=True, timeout=900, env=e, cwd=ROOT)
    tail = out.stdout[-1500:] + out.stderr[-500:]
    assert out.returncode == 0, tail
    assert " passed" in out.stdout and "failed" not in out.stdout, tail
