"""CPU mirror of the index arithmetic of the round-4 prefill GEMM (gpullama3.java_amd/csrc/gl3_prefill_gemm2.h): the LDS-DMA pieces of a K
stage must cover every 16-byte piece of the staged weight / activation image exactly once, land where the MFMA operand fetch reads them
(`la`, `lb` offsets), and the scale-operand identities the kernel rests on must hold in f32 / bf16 arithmetic — the three facts the GPU probe
(scripts/probes/scale_mfma_probe.hip) checks on the matrix pipe are re-derived here with NumPy so that the construction is pinned without a GPU."""
import itertools

import numpy as np

KB, TOK, TILE_BYTES = 2, 128, 2176


def dma_pieces(arows, nw):
    """(load id j, lane) -> ('A', c, row) or ('B', c, token) and its LDS byte offset inside the stage, as dma_off / dma_one compute them."""
    nla, nlb = KB * 2 * arows // 64, KB * 2 * TOK // 64
    off_bq = 2 * (KB * 2 * arows * 16)
    out = []
    for j in range(nla + nlb):
        for lane in range(64):
            if j < nla:
                e = 64 * j + lane
                c, row = divmod(e, arows)
                out.append((("A", c, row), 1024 * j + 16 * lane))
            else:
                jb = j - nla
                c = jb // (TOK // 64)
                tk = ((jb % (TOK // 64)) * 64 + lane) ^ c
                out.append((("B", c, tk), off_bq + 1024 * jb + 16 * lane))
    return out


def reader_offsets(arows):
    """LDS byte offsets the operand fetch uses: weights la + blk * (2 AROWS 16), activations OFF_BQ + ((b 2 + hi) TOK + (tk ^ (b 2 + hi))) 16."""
    off_bq = 2 * (KB * 2 * arows * 16)
    a = {("A", blk * 2 + hi, row): (hi * arows + row) * 16 + blk * (2 * arows * 16) for blk in range(KB) for hi in range(2) for row in range(arows)}
    b = {("B", blk * 2 + hi, tk): off_bq + ((blk * 2 + hi) * TOK + (tk ^ (blk * 2 + hi))) * 16 for blk in range(KB) for hi in range(2) for tk in range(TOK)}
    return {**a, **b}


def test_every_piece_once_and_where_the_reader_looks():
    for arows, nw in ((128, 4), (64, 4), (64, 8)):
        pieces = dma_pieces(arows, nw)
        want = reader_offsets(arows)
        assert len(pieces) == len(want) == KB * 2 * (arows + TOK)
        seen = {}
        for key, off in pieces:
            assert key not in seen, key
            seen[key] = off
            assert want[key] == off, (arows, key, off, want[key])
        # the xor spread keeps a 16-lane read group on 16 different 16-byte slots (conflict-free ds_read_b128)
        for c in range(2 * KB):
            slots = sorted((tk ^ c) for tk in range(16))
            assert slots == list(range(16))


def test_source_offsets_stay_inside_the_q8t_tile():
    # piece (c = blk * 2 + half, row): byte 128 (+1024 for the high half) + 16 * (lane of (blk, row & 15)) of the strip's tile, + 512 for odd stages
    for kb, c, r in itertools.product(range(4), range(4), range(16)):
        lane = (kb & 1) * 32 + (c >> 1) * 16 + r
        off = (1152 if c & 1 else 128) + 16 * lane
        assert 128 <= off and off + 16 <= TILE_BYTES
        blk_in_group = lane >> 4
        assert blk_in_group == (kb & 1) * 2 + (c >> 1)           # stage kb holds blocks 2 kb, 2 kb + 1 of tile group kb >> 1


def f16_values(rng, n):
    bits = rng.integers(0, 0x7C00, n, dtype=np.uint16)
    bits[: n // 8] = rng.integers(0, 1024, n // 8, dtype=np.uint16)          # subnormals and zero
    bits[n // 8] = 0x7BFF
    return bits.view(np.float16).astype(np.float32)


def bf16_split(x):
    hi = (x.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
    lo = x - hi
    assert np.all((lo.view(np.uint32) & np.uint32(0xFFFF)) == 0), "the low part must be a bf16 value too"
    return hi, lo


def test_scale_product_identities_in_numpy():
    rng = np.random.default_rng(3)
    w = f16_values(rng, 4096) * np.where(rng.random(4096) < 0.25, -1.0, 1.0).astype(np.float32)      # a GGUF may carry negative block scales
    a = f16_values(rng, 4096)
    s = w * a                                                     # f32 product
    assert np.array_equal(s.astype(np.float64), w.astype(np.float64) * a.astype(np.float64)), "s = wScale * aScale is exact in f32"
    B = np.float32(12582912.0)
    nbs = -B * s
    assert np.array_equal(nbs.astype(np.float64), -12582912.0 * s.astype(np.float64)), "B s is exact in f32"
    whi, wlo = bf16_split(w)
    ahi, alo = bf16_split(a)
    # the four-term sum, in any order, in float64 (exact) equals s; every partial sum fits 24 bits
    terms = [whi * ahi, whi * alo, wlo * ahi, wlo * alo]
    for perm in itertools.permutations(range(4)):
        acc = np.zeros_like(s)
        for t in perm:
            acc = acc + terms[t]                                  # f32 adds: exact because every partial sum is representable
        assert np.array_equal(acc, s), perm
    isum = rng.integers(-516128, 516129, 4096).astype(np.int32)
    D = (np.int32(0x4B400000) + isum).view(np.float32)            # the int8 MFMA's biased accumulator read as f32
    assert np.array_equal(D.astype(np.float64), 12582912.0 + isum)
    fused = (D.astype(np.float64) * s.astype(np.float64) + nbs.astype(np.float64)).astype(np.float32)    # fma: one rounding of the exact value
    ref = isum.astype(np.float32) * s                             # the reference: float(isum) * (wScale * aScale)
    same = (fused == ref) | ((fused == 0) & (ref == 0))           # +-0 differ in sign only; acc + (+-0) is the same value
    assert np.all(same)
