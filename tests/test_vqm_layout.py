"""Mirror of the host / index arithmetic of gemm_vlq_mfma_kernel (gpullama3.java_amd/csrc/gl3_prefill_vl.h): the XCD-pinned tile mapping
(vqm_groups / vqm_grid / vqm_tile_of) must hand every (row tile, token tile) of a launch to exactly one workgroup for every tile count a
model or a tensor-parallel rank can present, and the f32 quant layout of a K stage (index b * SB + l * SL + q * SQ + row) must be a
bijection into the stage's LDS slice whose staging writes and operand reads are bank-conflict-free.  CPU only; the GPU parity tests run a
handful of these shapes."""
import itertools

VQM_ROWS, VQM_TOK, VQM_XP, VQM_SL, VQM_SQ = 64, 64, 68, 68, 560
VQM_SB = 4 * VQM_SQ
VQM_STAGE_FLOATS = VQM_TOK * VQM_XP + 2 * VQM_SB + 2 * VQM_ROWS


def vqm_groups(ntt):
    return 8 if ntt >= 8 else 4 if ntt >= 4 else 2 if ntt >= 2 else 1


def vqm_grid(nrt, ntt):
    G = vqm_groups(ntt); P = 8 // G
    return 8 * ((nrt + P - 1) // P) * ((ntt + G - 1) // G)


def vqm_tile_of(bid, nrt, ntt):
    G = vqm_groups(ntt); P = 8 // G; x = bid & 7; slot = bid >> 3
    nrp = (nrt + P - 1) // P
    rt = (x // G) + P * (slot % nrp)
    tt = (x % G) + G * (slot // nrp)
    return (rt, tt) if rt < nrt and tt < ntt else None


def test_every_tile_is_owned_by_exactly_one_workgroup():
    for nrt, ntt in itertools.product(list(range(1, 70)) + [96, 224, 448, 501, 2004], range(1, 34)):
        seen = [vqm_tile_of(b, nrt, ntt) for b in range(vqm_grid(nrt, ntt))]
        owned = [t for t in seen if t is not None]
        assert len(owned) == len(set(owned)) == nrt * ntt, (nrt, ntt)


def test_token_tiles_of_an_xcd_are_fixed_while_row_tiles_stream():
    """XCD x = workgroup id & 7 only ever sees token tiles congruent to x modulo the group count: its x tiles stay in that XCD's L2."""
    for nrt, ntt in ((224, 8), (64, 8), (96, 16), (224, 2), (448, 4), (7, 3)):
        G = vqm_groups(ntt)
        for b in range(vqm_grid(nrt, ntt)):
            t = vqm_tile_of(b, nrt, ntt)
            if t is not None:
                assert t[1] % G == (b & 7) % G


def test_quant_layout_is_a_bijection_inside_the_stage():
    idx = {b * VQM_SB + l * VQM_SL + q * VQM_SQ + row for b in range(2) for l in range(8) for q in range(4) for row in range(VQM_ROWS)}
    assert len(idx) == 2 * 8 * 4 * VQM_ROWS
    assert max(idx) < 2 * VQM_SB
    assert VQM_STAGE_FLOATS * 4 * 2 <= 80 * 1024          # two workgroups per CU (160 KB of LDS)


def banks(addresses):
    return [a % 32 for a in addresses]


def test_lds_accesses_are_bank_conflict_free():
    # staging write: thread (wave w, rr, l) writes index l * SL + w * 8 + rr (+ b, q terms common to the wavefront); 32 lanes per cycle
    for half in range(2):
        lanes = range(32 * half, 32 * half + 32)
        assert len(set(banks([(ln & 7) * VQM_SL + (ln >> 3) for ln in lanes]))) == 32
    # operand read of the quants: lane (q = lane >> 4, row = lane & 15) reads q * SQ + row (+ common terms)
    for half in range(2):
        lanes = range(32 * half, 32 * half + 32)
        assert len(set(banks([(ln >> 4) * VQM_SQ + (ln & 15) for ln in lanes]))) == 32
    # operand read of x: ds_read_b128, lane (q, token i) reads 4 floats at i * XP + 8 q; 8 lanes per cycle must cover 32 distinct banks
    for first in range(0, 64, 8):
        got = []
        for ln in range(first, first + 8):
            base = (ln & 15) * VQM_XP + 8 * (ln >> 4)
            got += banks(range(base, base + 4))
        assert len(set(got)) == 32, first
