"""CPU mirrors of the index arithmetic of the pinned LDS-read rings (r4): every element is consumed exactly once, in ascending order,
and every clamped refill stays inside the row.

* seq_sum_lds_ring (gl3_decode_kernels.h): groups of 16, three groups in flight, two tail groups, then quads, then single elements;
* the weighted-V loop of attn_head_kernel: groups of 4 timesteps, three in flight, then single timesteps;
* the Qwen2-MoE router chain (gl3_moe_kernels.h): groups of 16 over a row whose length is a multiple of 32;
* the slot -> (matrix, rows) map of the routed experts' launch with the shared expert's chunks as extra slots (MoeSlots).
"""
import pytest


def _ring(n_groups, group, consume, load):
    """The common loop shape: load 0, 1, 2; per trip use A / refill A (g + 3), use B / refill B (g + 4), use C / refill C (g + 5)."""
    held = {}
    for r, g in zip("abc", (0, 1, 2)):
        held[r] = load(min(g, n_groups - 1))
    g = 0
    while g + 3 <= n_groups:
        for k, r in enumerate("abc"):
            consume(held[r])
            held[r] = load(min(g + 3 + k, n_groups - 1))
        g += 3
    for r in "ab":
        if g < n_groups:
            consume(held[r])
            g += 1
    return g * group


@pytest.mark.parametrize("n", list(range(0, 70)) + [127, 128, 129, 255, 256, 511, 512, 1000, 2048, 16384])
def test_seq_sum_ring_consumes_every_element_once_in_order(n):
    order, loads = [], []
    G = n >> 4
    i = 0
    if G >= 3:
        def load(g):
            loads.append((16 * g, 16 * g + 16))
            return 16 * g
        i = _ring(G, 16, lambda base: order.extend(range(base, base + 16)), load)
        assert i == 16 * G
    while i + 4 <= n:
        order.extend(range(i, i + 4))
        i += 4
    order.extend(range(i, n))
    assert order == list(range(n))
    assert all(0 <= lo and hi <= n for lo, hi in loads)                  # clamped refills re-read the last group, never past the row


@pytest.mark.parametrize("n", list(range(1, 40)) + [63, 64, 65, 100, 127, 128])
def test_weighted_v_ring_consumes_every_timestep_once_in_order(n):
    """attn_head_kernel: positions < AF_MAXN = 128, n = pos + 1."""
    order, loads = [], []
    NG = n >> 2
    tt = 0
    if NG >= 3:
        def load(g):
            loads.append(4 * g + 3)
            return 4 * g
        tt = _ring(NG, 4, lambda base: order.extend(range(base, base + 4)), load)
        assert tt == 4 * NG
    order.extend(range(tt, n))
    assert order == list(range(n))
    assert all(hi < n for hi in loads)


@pytest.mark.parametrize("dim", [32, 64, 96, 256, 1024, 2048, 2560, 4096, 4480])
def test_moe_router_chain_covers_the_row(dim):
    """gl3_moe_kernels.h: G = dim / 16 groups (dim is a multiple of 32, so G is even and at least 2)."""
    assert dim % 32 == 0
    order = []
    G = dim >> 4
    # the kernel's ring has no "G >= 3" guard: the initial loads clamp and the two tail steps take what is left (G = 2 for dim 32 / 64)
    done = _ring(G, 16, lambda base: order.extend(range(base, base + 16)), lambda g: 16 * g)
    assert done == dim and order == list(range(dim))


@pytest.mark.parametrize("topk,mh,shared", [(4, 1408, 5632), (2, 128, 512), (4, 384, 1536), (4, 1408, 5000), (8, 256, 256), (4, 512, 100)])
def test_moe_slots_cover_the_selected_experts_and_the_shared_expert(topk, mh, shared):
    """launch_matvec_sel / MoeSlots: slots j < topk work on expert sel[j] (rows [0, mh) of its sub-matrix, output slot j); the following
    ceil(shared / mh) slots are mh-row chunks of the shared expert's dense matrices, the last one possibly shorter."""
    n_slots = topk + (shared + mh - 1) // mh
    covered = []
    for slot in range(n_slots):
        if slot < topk:
            rows, strips = mh, mh // 16
            assert rows == 16 * strips or mh % 16
        else:
            c = slot - topk
            left = shared - c * mh
            rows = min(mh, left)
            assert rows > 0
            covered.extend(range(c * mh, c * mh + rows))
            assert (rows + 15) // 16 <= (mh + 15) // 16              # a chunk never needs more strips than the launch's grid.x covers
    assert covered == list(range(shared))
