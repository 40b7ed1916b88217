"""CPU check of the arithmetic identities the Q4_0 kernels' nibble unpack rests on (gl3_veclane_kernels.h, q4_plane / q4_mul):
(1) (n | 0x6400) read as IEEE f16 is 1024 + n for every nibble n, and subtracting 1032 in f16 gives exactly n - 8;
(2) fma(x, h, -0.0) with h = n - 8 equals the f32 product x * h (one rounding), including the sign of a zero product;
(3) the chunked strictly sequential f32 sum with a carried start value equals the one-pass sequential sum (windowed softmax rows).
Pure NumPy, no GPU, no library."""
import numpy as np


def test_f16_magic_nibbles_are_exact():
    for n in range(16):
        h = np.array([0x6400 | n], dtype=np.uint16).view(np.float16)[0]
        assert float(h) == 1024.0 + n
        d = np.float16(h) + np.float16(-1032.0)                 # v_pk_add_f16: one f16 rounding, exact here
        assert float(d) == float(n - 8)
    # the packed form: two nibbles 16 bits apart in a dword
    w = np.uint32(0xA3C59F07)
    for shift in (0, 4, 8, 12):
        bits = ((w >> np.uint32(shift)) & np.uint32(0x000F000F)) | np.uint32(0x64006400)
        lo, hi = np.array([bits & 0xFFFF], dtype=np.uint16).view(np.float16)[0], np.array([bits >> 16], dtype=np.uint16).view(np.float16)[0]
        assert float(lo) - 1032.0 == float(((int(w) >> shift) & 0xF) - 8)
        assert float(hi) - 1032.0 == float(((int(w) >> (shift + 16)) & 0xF) - 8)


def test_fma_with_negative_zero_addend_is_the_rounded_product():
    rng = np.random.default_rng(3)
    xs = np.concatenate([rng.standard_normal(4000).astype(np.float32) * np.float32(3.7),
                         np.array([0.0, -0.0, 1e-40, -1e-40, 3.4e38, -3.4e38, 1.17549435e-38], dtype=np.float32)])
    for n in range(16):
        h = np.float32(n - 8)
        with np.errstate(over="ignore"):                         # +-3.4e38 * 8 overflows to inf on both sides
            prod = xs * h                                        # f32 multiply, one rounding
            fma = (xs.astype(np.float64) * np.float64(h) + np.float64(-0.0)).astype(np.float32)   # exact product (<= 28 bits) + (-0), one rounding
        assert np.array_equal(prod.view(np.uint32), fma.view(np.uint32)), n


def test_chunked_sequential_sum_carries_exactly():
    rng = np.random.default_rng(5)
    e = np.exp(rng.standard_normal(5000).astype(np.float32) * np.float32(2.0)).astype(np.float32)

    def seq(v, start=np.float32(0.0)):
        s = np.float32(start)
        for x in v:
            s = np.float32(s + x)
        return s

    one = seq(e)
    for w in (1024, 2048, 4096):
        s = np.float32(0.0)
        for c0 in range(0, len(e), w):
            s = seq(e[c0:c0 + w], s)
        assert s.view(np.uint32) == one.view(np.uint32)
