"""K-quant -> Q8_0 load-time conversion (ModelLoader.dequantizeToQ8_0TornadoTensor,
J/model/loader/ModelLoader.java:173-224): the native converter (gl3_kquant_to_q8_0, per-element getFloat as the reference's
Q4_K / Q5_K / Q6_K FloatTensor classes) against an independent NumPy restatement written from the ggml block format."""
import ctypes as C

import numpy as np
import pytest

import kquant_np as kq


def _convert(hip, ggml_type, raw, n):
    out = np.empty(n // 32 * 34, np.uint8)
    rc = hip.lib().gl3_kquant_to_q8_0(ggml_type, raw.ctypes.data_as(C.c_void_p), n, out.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return out


@pytest.mark.parametrize("ggml_type", [12, 13, 14])
def test_native_kquant_conversion_matches_numpy_restatement(pkg, ggml_type):
    from importlib import import_module
    import __graft_entry__ as ge
    hip = import_module(ge.PKG_NAME + ".hip")
    rng = np.random.default_rng(100 + ggml_type)
    n = 256 * 257
    raw = kq.random_blocks(ggml_type, n, rng)
    want = kq.to_q8_0(kq.DEQUANT[ggml_type](raw, n))
    got = _convert(hip, ggml_type, raw, n)
    assert np.array_equal(got, want)
    # all-zero super-block: scale 0 -> q = 0 (invScale = 0 branch)
    z = np.zeros(raw.size // (n // 256), np.uint8)
    assert not _convert(hip, ggml_type, z, 256).any()
    assert hip.lib().gl3_kquant_to_q8_0(8, raw.ctypes.data_as(C.c_void_p), n, got.ctypes.data_as(C.c_void_p)) == -2     # not a K-quant
    assert hip.lib().gl3_kquant_to_q8_0(ggml_type, raw.ctypes.data_as(C.c_void_p), 100, got.ctypes.data_as(C.c_void_p)) == -1


def test_hand_built_q4_k_block():
    """One Q4_K super-block by hand: d = 1, dmin = 0.5, sub-block j: scale j + 1, min j; nibbles = position mod 16."""
    b = np.zeros(144, np.uint8)
    b[0:2] = np.array([1.0], np.float16).view(np.uint8)
    b[2:4] = np.array([0.5], np.float16).view(np.uint8)
    sc = [j + 1 for j in range(8)]
    mn = list(range(8))
    for j in range(4):
        b[4 + j] = sc[j] | ((sc[j + 4] >> 4) << 6)
        b[8 + j] = mn[j] | ((mn[j + 4] >> 4) << 6)
        b[12 + j] = (sc[j + 4] & 0xF) | ((mn[j + 4] & 0xF) << 4)
    for p in range(4):
        for i in range(32):
            b[16 + p * 32 + i] = (i % 16) | (((i + 1) % 16) << 4)
    x = kq.dequant_q4_k(b, 256)
    for j in range(8):
        for i in range(32):
            q = (i % 16) if j % 2 == 0 else ((i + 1) % 16)
            assert x[j * 32 + i] == np.float32(1.0 * sc[j] * q - 0.5 * mn[j])
