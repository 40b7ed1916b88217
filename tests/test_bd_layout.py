"""Small-batch operand layout (csrc/gl3_bd_gemm.h, bdq_offset / bds_offset in csrc/gl3_decode_kernels.h) restated in Python:
the quantiser's scatter must be a bijection onto the [pair][token slot][64 B] / [tile][token slot][4] images, and the byte
order inside a 64-byte pair must be the one the weight side's v_permlane32_swap produces (k-group g of the 16x16x32 int8 MFMA
holds k-chunk 0, 2, 1, 3 of its block: k 0-7, 16-23, 8-15, 24-31) — the GPU parity tests check the values, this checks the map."""
import itertools


def bdq_offset(qd, tok, tslots):
    blk, qi = qd >> 3, qd & 7
    c = qi >> 1
    g = ((c & 1) << 1) | (c >> 1)
    return ((blk >> 1) * tslots + tok) * 64 + 16 * g + 8 * (blk & 1) + 4 * (qi & 1)


def bds_offset(blk, tok, tslots):
    return ((blk >> 2) * tslots + tok) * 4 + (blk & 3)


def test_quantiser_scatter_is_a_bijection():
    tslots, k = 32, 2560
    seen = set()
    for tok, qd in itertools.product(range(tslots), range(k // 4)):
        off = bdq_offset(qd, tok, tslots)
        assert off % 4 == 0 and off not in seen
        seen.add(off)
    assert seen == set(range(0, (k // 64) * tslots * 64, 4))
    sc = {bds_offset(b, tok, tslots) for tok in range(tslots) for b in range(k // 32)}
    assert sc == set(range((k // 128) * tslots * 4))


def test_lane_operand_order_matches_the_swapped_weight_side():
    # lane (token t, k-group g) loads 16 B at pair_base + t * 64 + 16 g: low 8 B feed the MFMA of block 2j, high 8 B that of 2j + 1
    chunk_of_group = {0: 0, 1: 2, 2: 1, 3: 3}            # after the swap: g = 0..3 hold k 0-7, 16-23, 8-15, 24-31
    for blk in (0, 1, 6, 7):
        for g in range(4):
            c = chunk_of_group[g]
            for half in range(2):                        # the two quads (4 k each) of an 8-k chunk
                qd = blk * 8 + 2 * c + half
                off = bdq_offset(qd, 5, 32)
                assert off == ((blk >> 1) * 32 + 5) * 64 + 16 * g + 8 * (blk & 1) + 4 * half
