"""TEST HELPER — NumPy restatement of the K-quant dequantisation (ggml dequantize_row_q4_K / q5_K / q6_K structure, with the f32
operation order of the reference's getFloat: (d * sc) * q - (dmin * m); (d * sc) * q for Q6_K) and of
ModelLoader.dequantizeToQ8_0TornadoTensor (J/model/loader/ModelLoader.java:173-224).  Written from the format, independently of
the per-element C++ in csrc/gl3_gguf.cpp; used by tests/test_kquant.py and tests/test_gpu_kquant.py."""
import numpy as np

F32 = np.float32


def _scale_min_k4(sc):          # sc: [nb, 12] uint8 -> scales [nb, 8], mins [nb, 8] (6-bit values)
    sc = sc.astype(np.int32)
    s = np.empty(sc.shape[:-1] + (8,), np.int32)
    m = np.empty_like(s)
    s[..., :4] = sc[..., 0:4] & 63
    m[..., :4] = sc[..., 4:8] & 63
    s[..., 4:] = (sc[..., 8:12] & 0xF) | ((sc[..., 0:4] >> 6) << 4)
    m[..., 4:] = (sc[..., 8:12] >> 4) | ((sc[..., 4:8] >> 6) << 4)
    return s, m


def dequant_q4_k(raw, n):
    b = np.frombuffer(raw, np.uint8)[: n // 256 * 144].reshape(-1, 144)
    d = b[:, 0:2].copy().view(np.float16).astype(F32)          # [nb, 1]
    dmin = b[:, 2:4].copy().view(np.float16).astype(F32)
    s, m = _scale_min_k4(b[:, 4:16])
    qs = b[:, 16:144].reshape(-1, 4, 32)
    q = np.stack([qs & 0xF, qs >> 4], axis=2).reshape(-1, 8, 32).astype(F32)       # sub-block 2p = low nibbles, 2p+1 = high
    out = (d[:, :, None] * s.astype(F32)[:, :, None]).astype(F32) * q - (dmin[:, :, None] * m.astype(F32)[:, :, None]).astype(F32)
    return out.astype(F32).reshape(-1)


def dequant_q5_k(raw, n):
    b = np.frombuffer(raw, np.uint8)[: n // 256 * 176].reshape(-1, 176)
    d = b[:, 0:2].copy().view(np.float16).astype(F32)
    dmin = b[:, 2:4].copy().view(np.float16).astype(F32)
    s, m = _scale_min_k4(b[:, 4:16])
    qh = b[:, 16:48]                                              # [nb, 32]
    qs = b[:, 48:176].reshape(-1, 4, 32)
    lo, hi = (qs & 0xF).astype(np.int32), (qs >> 4).astype(np.int32)
    for p in range(4):
        lo[:, p] += ((qh >> (2 * p)) & 1).astype(np.int32) * 16
        hi[:, p] += ((qh >> (2 * p + 1)) & 1).astype(np.int32) * 16
    q = np.stack([lo, hi], axis=2).reshape(-1, 8, 32).astype(F32)
    out = (d[:, :, None] * s.astype(F32)[:, :, None]).astype(F32) * q - (dmin[:, :, None] * m.astype(F32)[:, :, None]).astype(F32)
    return out.astype(F32).reshape(-1)


def dequant_q6_k(raw, n):
    b = np.frombuffer(raw, np.uint8)[: n // 256 * 210].reshape(-1, 210)
    ql = b[:, 0:128].reshape(-1, 2, 64).astype(np.int32)
    qh = b[:, 128:192].reshape(-1, 2, 32).astype(np.int32)
    sc = b[:, 192:208].copy().view(np.int8).reshape(-1, 2, 8).astype(F32)
    d = b[:, 208:210].copy().view(np.float16).astype(F32)          # [nb, 1]
    q = np.empty((b.shape[0], 2, 4, 32), np.int32)
    q[:, :, 0] = (ql[:, :, 0:32] & 0xF) | (((qh >> 0) & 3) << 4)
    q[:, :, 1] = (ql[:, :, 32:64] & 0xF) | (((qh >> 2) & 3) << 4)
    q[:, :, 2] = (ql[:, :, 0:32] >> 4) | (((qh >> 4) & 3) << 4)
    q[:, :, 3] = (ql[:, :, 32:64] >> 4) | (((qh >> 6) & 3) << 4)
    q = (q - 32).astype(F32)                                       # [nb, half, group, 32]
    scale = sc.reshape(-1, 2, 4, 2)                                # scale index = 2 * group + (pos // 16)
    scale = np.repeat(scale, 16, axis=3)                           # [nb, half, group, 32]
    out = (d[:, :, None, None] * scale).astype(F32) * q
    return out.astype(F32).reshape(-1)


DEQUANT = {12: dequant_q4_k, 13: dequant_q5_k, 14: dequant_q6_k}


def to_q8_0(x):
    """dequantizeToQ8_0TornadoTensor on an f32 vector (multiple of 32)."""
    xb = np.asarray(x, F32).reshape(-1, 32)
    max_abs = np.max(np.abs(xb), axis=1).astype(F32)
    scale = (max_abs / F32(127.0)).astype(F32)
    with np.errstate(divide="ignore"):
        inv = np.where(scale != 0, F32(1.0) / scale, F32(0)).astype(F32)
    q = np.floor((xb * inv[:, None]).astype(F32).astype(np.float64) + 0.5)       # Math.round(float): floor(a + 1/2), the sum exact
    q = np.clip(q, -128, 127).astype(np.int8)
    out = np.empty((xb.shape[0], 34), np.uint8)
    out[:, :2] = scale.astype(np.float16).view(np.uint8).reshape(-1, 2)
    out[:, 2:] = q.view(np.uint8)
    return out.reshape(-1)


def random_blocks(ggml_type, n, rng, scale=2e-3):
    """n elements of random K-quant super-blocks with finite, model-like f16 scales."""
    bs = {12: 144, 13: 176, 14: 210}[ggml_type]
    nb = n // 256
    b = rng.integers(0, 256, (nb, bs), dtype=np.uint8)
    d = (np.abs(rng.standard_normal(nb)) * scale + scale * 0.1).astype(np.float16).view(np.uint8).reshape(nb, 2)
    if ggml_type == 14:
        b[:, 208:210] = d
    else:
        b[:, 0:2] = d
        b[:, 2:4] = (np.abs(rng.standard_normal(nb)) * scale * 4).astype(np.float16).view(np.uint8).reshape(nb, 2)
    return b.reshape(-1)
