"""Mirror of the K-split decode matvec's work partition (gpullama3.java_amd/csrc/gl3_veclane_kernels.h: vq_round_chunks, vq_waves,
and the kernel's nrounds / rc / c_lo / nc arithmetic): for every (weight type, K, matrices per launch, 8-row groups) a model or a
tensor-parallel rank can present, the wavefronts' chunk ranges must tile [0, K / chunk) exactly once, in K order across
(round, wavefront) — the order in which the fma chain is handed from wavefront to wavefront — and fit the per-round register
budget and the LDS request.  CPU only; the GPU parity tests exercise a handful of these shapes."""
import itertools

Q4_0, Q8_0 = 2, 3


def round_chunks(wt, maxw, nm):
    """vq_round_chunks<WT, MAXW>(nm): chunks per wavefront and round (integer divisions in the kernel's order)."""
    return (8 if wt == Q4_0 else 12) // (2 if maxw == 16 else 1) // nm


def x_quads(wt, maxw, nm):
    """XQ of matvec_vlq_kernel: float4 loads per lane that stage a round's activation slice (rounded up)."""
    return (round_chunks(wt, maxw, nm) * chunk_elems(wt) + 255) // 256


def chunk_elems(wt):
    return 256 if wt == Q4_0 else 128


def vq_waves(wt, k, nm, ngroups):
    nch = k // chunk_elems(wt)
    nw = 4
    while nw < 16 and ngroups * nw < 2048:
        nw *= 2
    if nw == 4 and nch > 4 * round_chunks(wt, 8, nm):
        nw = 8
    while nw > 4 and nch < nw:
        nw //= 2
    return nw


def kernel_ranges(wt, k, nm, nw):
    maxw = 16 if nw == 16 else 8
    rcm = round_chunks(wt, maxw, nm)
    nch = k // chunk_elems(wt)
    nrounds = (nch + nw * rcm - 1) // (nw * rcm)
    rc = (nch + nw * nrounds - 1) // (nw * nrounds)
    assert 1 <= rc <= rcm, (wt, k, nm, nw, rc, rcm)
    # the x staging registers (XQ float4 per lane, 64 lanes) must cover every chunk of a round (r3 advisor finding: Q8_0 /
    # two matrices / 16 wavefronts has rcm * 128 = 384 floats, and a truncated XQ = 1 staged only 256 of them)
    assert rc * chunk_elems(wt) <= x_quads(wt, maxw, nm) * 256, (wt, k, nm, nw, rc)
    order = []
    for r in range(nrounds):
        for w in range(nw):                      # chain order inside a round: wavefront 0, 1, ...
            c_lo = (r * nw + w) * rc
            nc = max(0, min(rc, nch - c_lo))
            order.extend(range(c_lo, c_lo + nc))
            # the loads are unconditional with clamped indices: they must stay inside the matrix
            c_lo_l = min(c_lo, nch - 1)
            nc_l = max(1, min(rc, nch - c_lo_l))
            assert 0 <= c_lo_l and c_lo_l + nc_l <= nch
    lds_floats = nm * 64 + nw * rc * chunk_elems(wt)
    return order, nch, lds_floats


def test_every_chunk_once_in_k_order():
    ks = [256, 512, 768, 1024, 2048, 2560, 3072, 4096, 5120, 8192, 9728, 11008, 12288, 14336, 28672]
    for wt, k, nm in itertools.product((Q4_0, Q8_0), ks, (1, 2)):
        if k % chunk_elems(wt):
            continue
        for ngroups in (1, 8, 24, 64, 96, 224, 448, 512, 768, 1792, 16032):
            nw = vq_waves(wt, k, nm, ngroups)
            assert nw in (4, 8, 16)
            order, nch, lds = kernel_ranges(wt, k, nm, nw)
            assert order == list(range(nch)), (wt, k, nm, ngroups, nw)
            assert lds * 4 <= (nm * 64 + k + nw * chunk_elems(wt)) * 4      # the launcher's LDS bound
        for nw in (4, 8, 16):                                                 # any wave count the kernel can be launched with
            order, nch, _ = kernel_ranges(wt, k, nm, nw)
            assert order == list(range(nch)), (wt, k, nm, nw)


def test_wave_count_fills_the_chip_for_rank_slices():
    # Llama-3-8B Q4_0 residual projections on one GPU and on a tp = 8 rank (rows / 8)
    assert vq_waves(Q4_0, 4096, 1, 512) == 4          # wo: 512 groups x 4 = 2048 wavefronts
    assert vq_waves(Q4_0, 14336, 1, 512) == 8         # down: one round must cover 56 chunks
    assert vq_waves(Q4_0, 4096, 1, 64) == 16          # wo slice of a tp = 8 rank
    assert vq_waves(Q4_0, 14336, 1, 64) == 16
    assert vq_waves(Q4_0, 256, 1, 8) == 4             # never more wavefronts than chunks (beyond the minimum of 4)


def test_round_chunks_match_the_kernel_formula():
    assert [round_chunks(Q4_0, 8, 1), round_chunks(Q4_0, 16, 1), round_chunks(Q4_0, 8, 2), round_chunks(Q4_0, 16, 2)] == [8, 4, 4, 2]
    assert [round_chunks(Q8_0, 8, 1), round_chunks(Q8_0, 16, 1), round_chunks(Q8_0, 8, 2), round_chunks(Q8_0, 16, 2)] == [12, 6, 6, 3]
    # the shape of the finding: dim 5120 / hidden 6912 at tp = 8 -> 108 groups, 16 wavefronts, 40 chunks, rc = 3
    assert vq_waves(Q8_0, 5120, 2, 108) == 16
    _, nch, _ = kernel_ranges(Q8_0, 5120, 2, 16)
    assert nch == 40 and x_quads(Q8_0, 16, 2) == 2
