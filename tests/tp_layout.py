"""TEST HELPER — tensor-parallel partition of the forward pass (a Python description of what gl3_create / gl3_upload_tensor do,
used only by tests/test_tp_gloo.py to check the scheme on CPU over gloo; the product code is csrc/gl3_api.hip).

Every matrix except Wo is split by OUTPUT rows so that each dot product stays whole and in the reference's order on one
rank (bit-identical results); activations are re-assembled with all-gathers.  Wo (attn_output) is REPLICATED: every rank
computes the whole projection from the gathered attention output, so no gather follows it (3 per layer: xb, hb, x).  The same table drives the C++
upload slicing (csrc/gl3_api.hip: gl3_upload_tensor) and the CPU test of the scheme (tests/test_tp_gloo.py).
"""
from __future__ import annotations


def validate(cfg, tp: int) -> None:
    """Mirrors the checks of gl3_create (GL3_E_UNSUPPORTED)."""
    if tp < 1:
        raise ValueError("tp_size must be >= 1")
    if cfg.n_heads % tp or cfg.n_kv_heads % tp or cfg.hidden % (16 * tp) or cfg.vocab % (16 * tp) or cfg.dim % (16 * tp):
        raise ValueError("tp_size must divide n_heads, n_kv_heads, hidden/16, dim/16 and vocab/16")


def row_slices(cfg, tp: int, rank: int) -> dict:
    """GGUF tensor suffix -> (first row, number of rows) kept by `rank`."""
    validate(cfg, tp)
    hs = cfg.head_size
    ql, kvl = cfg.n_heads // tp * hs, cfg.n_kv_heads // tp * hs
    hl, dl, vl = cfg.hidden // tp, cfg.dim // tp, cfg.vocab // tp
    return {
        "attn_q.weight": (rank * ql, ql), "attn_k.weight": (rank * kvl, kvl), "attn_v.weight": (rank * kvl, kvl),
        "attn_output.weight": (0, cfg.dim),        # replicated
        "ffn_gate.weight": (rank * hl, hl), "ffn_up.weight": (rank * hl, hl),
        "ffn_down.weight": (rank * dl, dl), "output.weight": (rank * vl, vl),
    }


# all-gather points of one layer, in order: (buffer, floats per rank)
def gather_points(cfg, tp: int):
    hs = cfg.head_size
    return [("xb", cfg.n_heads // tp * hs), ("hb", cfg.hidden // tp), ("x", cfg.dim // tp)]


# ---- batched prefill: rank-chunked activations (csrc/gl3_prefill.hip `chunked`) ----------------------------------------
# A [ntok][cols] activation produced by row-split matrices is stored as [tp][ntok][cols/tp]: rank r's GEMM writes the
# contiguous chunk r, the all-gather is in place, consumers index element j of token b at
#     (j // cc) * ntok * cc + b * cc + j % cc,   cc = cols // tp.
def chunked_index(b: int, j: int, cc: int, ntok: int) -> int:
    return (j // cc * ntok + b) * cc + j % cc


def to_chunked(x, tp: int):
    """[ntok][cols] -> flat [tp][ntok][cols/tp] (NumPy)."""
    import numpy as np
    ntok, cols = x.shape
    cc = cols // tp
    return np.ascontiguousarray(x.reshape(ntok, tp, cc).transpose(1, 0, 2)).reshape(-1)


def from_chunked(flat, tp: int, ntok: int, cols: int):
    cc = cols // tp
    return flat.reshape(tp, ntok, cc).transpose(1, 0, 2).reshape(ntok, cols)


def prefill_gather_points(cfg, tp: int, ntok: int):
    """all-gathers of one prefill layer: (buffer, floats per rank)."""
    hs = cfg.head_size
    return [("AO", ntok * (cfg.n_heads // tp * hs)), ("HB", ntok * (cfg.hidden // tp)), ("X", ntok * (cfg.dim // tp))]
