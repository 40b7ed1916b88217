"""NumPy restatement of GPULlama3.java's pure-Java forward pass (second, independent oracle).

TEST INFRASTRUCTURE ONLY — never imported by the product path (gpullama3.java_amd/).
PARITY UNPINNED: the reference holds no golden vectors for this path; this file
and oracle/gl3_oracle.c are written independently from the Java source and must agree bit for
bit (tests/test_oracle_cross.py), plus the hand-derived KATs in tests/test_oracle_kat.py.

Citations are relative to /root/reference/src/main/java/org/beehive/gpullama3/ (J/).

Numeric rules: every float op rounds to binary32 (NumPy float32 ufuncs), no FMA; sequential
sums use ``np.add.accumulate`` (strict left-to-right in float32 — unlike ``np.sum`` it never
pairwise-reassociates); Math.exp/sqrt/pow/cos/sin are evaluated in float64 and cast.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32

GGML_F32, GGML_F16, GGML_Q4_0, GGML_Q8_0 = 0, 1, 2, 8
# block size / type size — J/tensor/GGMLType.java:5-21
BLOCK = {GGML_F32: (1, 4), GGML_F16: (1, 2), GGML_Q4_0: (32, 18), GGML_Q8_0: (32, 34)}


def seq_sum(a: np.ndarray, axis: int = -1) -> np.ndarray:
    """Strict left-to-right float32 sum along ``axis`` starting from 0f (0+a0 == a0 exactly)."""
    a = np.asarray(a, dtype=F32)
    return np.take(np.add.accumulate(a, axis=axis, dtype=F32), -1, axis=axis)


def dequant(raw: np.ndarray, ggml_type: int, n: int) -> np.ndarray:
    """FloatTensor.getFloat for every element — Q8_0FloatTensor.java:55-63, Q4_0FloatTensor.java:57-71,
    FP16FloatTensor.java:48-51."""
    raw = np.frombuffer(raw, dtype=np.uint8) if not isinstance(raw, np.ndarray) else raw.view(np.uint8).reshape(-1)
    if ggml_type == GGML_F32:
        return raw[: 4 * n].view(F32).copy()
    if ggml_type == GGML_F16:
        return raw[: 2 * n].view(np.float16).astype(F32)
    if ggml_type == GGML_Q8_0:
        blk = raw[: n // 32 * 34].reshape(-1, 34)
        d = blk[:, :2].copy().view(np.float16).astype(F32)          # [nb,1]
        q = blk[:, 2:].view(np.int8).astype(F32)                     # [nb,32]
        return (q * d).reshape(-1)
    if ggml_type == GGML_Q4_0:
        blk = raw[: n // 32 * 18].reshape(-1, 18)
        d = blk[:, :2].copy().view(np.float16).astype(F32)
        lo = (blk[:, 2:] & 0x0F).astype(np.int8) - 8
        hi = (blk[:, 2:] >> 4).astype(np.int8) - 8
        q = np.concatenate([lo, hi], axis=1).astype(F32)
        return (q * d).reshape(-1)
    raise ValueError(ggml_type)


def quantize_act(x: np.ndarray):
    """Activation side of dotQ8Activation — J/tensor/standard/Q8_0FloatTensor.java:90-123."""
    xb = np.asarray(x, dtype=F32).reshape(-1, 32)
    amax = np.max(np.abs(xb), axis=1).astype(F32)
    qs = amax / F32(127.0)
    ascale = qs.astype(np.float16).astype(F32)
    with np.errstate(divide="ignore"):
        ainv = np.where(qs != 0, F32(1.0) / qs, F32(0.0)).astype(F32)
    s = xb * ainv[:, None]
    aq = np.trunc(s + np.copysign(F32(0.5), s)).astype(np.int32)
    return aq, ascale


def matmul(w_raw, ggml_type: int, x: np.ndarray, d0: int, d1: int) -> np.ndarray:
    """FloatTensor.matmul (J/tensor/standard/FloatTensor.java:98-100) with the per-type dot."""
    x = np.asarray(x, dtype=F32)
    raw = w_raw.view(np.uint8).reshape(-1)
    if ggml_type == GGML_Q8_0:
        nb = d1 // 32
        blk = raw[: d0 * nb * 34].reshape(d0, nb, 34)
        wscale = blk[:, :, :2].copy().view(np.float16).astype(F32).reshape(d0, nb)
        wq = blk[:, :, 2:].view(np.int8).astype(np.int32)           # [d0,nb,32]
        aq, ascale = quantize_act(x)
        isum = np.einsum("rbi,bi->rb", wq, aq).astype(np.int32)      # exact int32
        prod = isum.astype(F32) * (wscale * ascale[None, :])          # isum * (wScale * aScale)
        return seq_sum(prod, axis=1)                                   # result += ..., b ascending
    # scalar mode: result += getFloat(j) * x[j]  — FloatTensor.java:86-92
    w = dequant(raw, ggml_type, d0 * d1).reshape(d0, d1)
    return seq_sum(w * x[None, :], axis=1)


def fma32(a: np.ndarray, b: np.ndarray, c: np.ndarray) -> np.ndarray:
    """Correctly rounded binary32 fused multiply-add (FloatVector.fma = Math.fma per lane), built from float64:
    the product of two float32 is exact in float64; the sum is rounded TO ODD (exact TwoSum error decides), and a
    round-to-odd result with >= 2 spare bits rounds to float32 like the infinitely precise value (Boldo & Melquiond)."""
    a64, b64, c64 = np.asarray(a, F32).astype(np.float64), np.asarray(b, F32).astype(np.float64), np.asarray(c, F32).astype(np.float64)
    p = a64 * b64
    s = p + c64
    bb = s - p
    err = (p - (s - bb)) + (c64 - bb)                      # exact: p + c64 = s + err
    odd = (s.view(np.int64) & 1).astype(bool)
    toward = np.where(err > 0, np.inf, -np.inf)
    s = np.where((err != 0) & ~odd, np.nextafter(s, toward), s)
    return s.astype(F32)


def f16_to_f32_daz(h: np.ndarray) -> np.ndarray:
    """FP16FloatTensor.vectorDot's f16 -> f32 bit trick (FP16FloatTensor.java:72-100): exact for normal values, subnormal
    weights become signed zero ("emulate DAZ")."""
    b = h.astype(np.uint32)
    mask = np.where((b & 0x7C00) != 0, np.uint32(0xFFFFFFFF), np.uint32(0))
    bits = ((b & 0x8000) << 16) | ((((b & 0x7FFF) + 0x1C000) << 13) & mask)
    return bits.astype(np.uint32).view(F32)


class UnsupportedSpecies(Exception):
    """The reference throws UnsupportedOperationException(F_SPECIES.toString()) — Q8_0FloatTensor.java:165-167, Q4_0FloatTensor.java:118-120:
    the Q8_0 / Q4_0 vector dots exist for 128- and 256-bit species only (a 512-bit host, e.g. AVX-512, needs -Dllama.VectorBitSize=256)."""


def matmul_vec(w_raw, ggml_type: int, x: np.ndarray, d0: int, d1: int, bits: int = 256) -> np.ndarray:
    """FloatTensor.matmul with the Vector-API dots of a `bits`-wide species, L = bits / 32 float lanes (FloatTensor.java:21-47):
      F16   FP16FloatTensor.vectorDot (FP16FloatTensor.java:63-110) is species-generic: val[l] = fma(w[i + l], x[i + l], val[l]), i += L;
      Q8_0  (f32 activation) Q8_0FloatTensor.vectorDot :125-175 — 256: one fma per block over four 8-lane products; 128: TWO fmas per
            block (bytes 0..15, then 16..31), each over four 4-lane products (:154-163); 512: throws (:165-167);
      Q4_0  Q4_0FloatTensor.vectorDot :82-133 — 256: lo nibbles = elements 0..15, hi = 16..31, four 8-lane products, one fma; 128: two
            fmas per block (lo bytes, then hi bytes), each over four 4-lane products (:107-117); 512: throws (:118-120).
    reduceLanes(ADD) in lane order from 0.  Sizes here are multiples of the block / lane count: the scalar tails are empty."""
    if bits not in (128, 256, 512):
        raise ValueError(bits)
    L = bits // 32
    x = np.asarray(x, dtype=F32)
    raw = w_raw.view(np.uint8).reshape(-1)
    val = np.zeros((d0, L), F32)
    if ggml_type == GGML_F16:
        assert d1 % L == 0
        w = f16_to_f32_daz(raw[: 2 * d0 * d1].view(np.uint16)).reshape(d0, d1 // L, L)
        xv = x.reshape(d1 // L, L)
        for i in range(d1 // L):
            val = fma32(w[:, i, :], xv[i][None, :], val)
        return seq_sum(val, axis=1)
    if bits == 512:
        raise UnsupportedSpecies("Species[float, 16, S_512_BIT]")
    nb = d1 // 32
    if ggml_type == GGML_Q4_0:
        blk = raw[: d0 * nb * 18].reshape(d0, nb, 18)
        ws = blk[:, :, :2].copy().view(np.float16).astype(F32).reshape(d0, nb)
        lo = ((blk[:, :, 2:] & 0x0F).astype(np.int8) - 8).astype(F32)          # [d0, nb, 16]: elements 0..15
        hi = ((blk[:, :, 2:] >> 4).astype(np.int8) - 8).astype(F32)            # elements 16..31
        q = np.concatenate([lo, hi], axis=2)                                   # [d0, nb, 32] in element order
    elif ggml_type == GGML_Q8_0:
        blk = raw[: d0 * nb * 34].reshape(d0, nb, 34)
        ws = blk[:, :, :2].copy().view(np.float16).astype(F32).reshape(d0, nb)
        q = blk[:, :, 2:].view(np.int8).astype(F32)                            # [d0, nb, 32]
    else:
        raise ValueError(ggml_type)
    # both types: element e of a block multiplies x[32 b + e]; one fma per group of 4 L elements (256: the block; 128: its halves),
    # sum_i = products of elements [i L, (i + 1) L) of the group, ((sum0 + sum1) + sum2) + sum3
    G = 4 * L
    xb = x.reshape(nb, 32 // G, 4, L)
    qg = q.reshape(d0, nb, 32 // G, 4, L)
    for b in range(nb):
        for h in range(32 // G):
            s0 = xb[b, h, 0][None, :] * qg[:, b, h, 0]
            s1 = xb[b, h, 1][None, :] * qg[:, b, h, 1]
            s2 = xb[b, h, 2][None, :] * qg[:, b, h, 2]
            s3 = xb[b, h, 3][None, :] * qg[:, b, h, 3]
            sm = ((s0 + s1) + s2) + s3
            val = fma32(sm, ws[:, b][:, None], val)
    return seq_sum(val, axis=1)


def matmul_v256(w_raw, ggml_type: int, x: np.ndarray, d0: int, d1: int) -> np.ndarray:
    return matmul_vec(w_raw, ggml_type, x, d0, d1, 256)


def rmsnorm(x: np.ndarray, w: np.ndarray, eps: float) -> np.ndarray:
    """InferenceCore.rmsnorm — J/inference/InferenceCore.java:39-48."""
    x = np.asarray(x, dtype=F32)
    ss = seq_sum(x * x)
    ss = F32(ss / F32(x.size))
    ss = F32(ss + F32(eps))
    ss = F32(1.0 / np.sqrt(np.float64(ss)))
    return (np.asarray(w, dtype=F32) * (ss * x)).astype(F32)


def softmax(a: np.ndarray) -> np.ndarray:
    """FloatTensor.softmaxInPlace — J/tensor/standard/FloatTensor.java:211-219."""
    a = np.asarray(a, dtype=F32)
    m = np.max(a)
    e = np.exp((a - m).astype(np.float64)).astype(F32)
    return (e / seq_sum(e)).astype(F32)


def rope_table(ctx: int, head_size: int, theta: float):
    """RoPE.precomputeFreqsCis, ropeScaling=false — J/inference/operation/RoPE.java:6-37."""
    i = np.arange(0, head_size, 2, dtype=np.float64)
    freq = (1.0 / np.power(np.float64(theta), i / np.float64(head_size))).astype(F32)
    val = (np.arange(ctx, dtype=F32)[:, None] * freq[None, :]).astype(F32)
    return np.cos(val.astype(np.float64)).astype(F32).reshape(-1), np.sin(val.astype(np.float64)).astype(F32).reshape(-1)


def rope_table_yarn(ctx: int, head_size: int, theta: float, factor: float, beta_fast: float, beta_slow: float,
                    log_multiplier: float, original_ctx: int):
    """RoPE.precomputeFreqsCisYaRN — J/inference/operation/RoPE.java:39-83 (Devstral 2).  Every Java float stays an F32 here."""
    f = F32
    theta32 = f(theta)

    def corr_dim(n_rot):
        ratio = f(f(original_ctx) / f(f(f(n_rot) * f(2.0)) * f(np.pi)))
        return f(f(f(head_size) * f(np.log(np.float64(ratio)))) / f(f(2.0) * f(np.log(np.float64(theta32)))))

    low, high = corr_dim(beta_fast), corr_dim(beta_slow)
    freq_scale = f(f(1.0) / f(factor))
    if log_multiplier > 0:
        mscale = f(f(1.0) + f(f(f(0.1) * f(log_multiplier)) * f(np.log(np.float64(f(f(1.0) / freq_scale))))))
    else:
        mscale = f(1.0)
    i = np.arange(0, head_size, 2, dtype=np.float64)
    extrap = (1.0 / np.power(np.float64(theta), i / np.float64(head_size))).astype(F32)
    interp = (freq_scale * extrap).astype(F32)
    span = max(f(0.001), f(high - low))
    y = ((np.arange(head_size // 2, dtype=F32) - low).astype(F32) / span).astype(F32)
    ramp = (f(1.0) - np.minimum(f(1.0), np.maximum(f(0.0), y))).astype(F32)
    freq = ((interp * (f(1.0) - ramp)).astype(F32) + (extrap * ramp).astype(F32)).astype(F32)
    val = (np.arange(ctx, dtype=F32)[:, None] * freq[None, :]).astype(F32)
    cr = (np.cos(val.astype(np.float64)).astype(F32) * mscale).astype(F32)
    ci = (np.sin(val.astype(np.float64)).astype(F32) * mscale).astype(F32)
    return cr.reshape(-1), ci.reshape(-1)


class NpOracle:
    """Holds config, raw GGUF-layout tensors and the State arrays (LlamaState.java:28-81)."""

    def __init__(self, cfg: dict, tensors: dict, rope, vector_bits: int = 0, f32_activation: bool = False):
        """vector_bits: 0 = scalar dots (-Dllama.VectorBitSize=0); 128 / 256 / 512 = Vector-API dots of that species for F16 / Q4_0 matrices
        (and Q8_0 with the f32 activation) — VectorShape.preferredShape() of the host: 256 on AVX2, 512 on AVX-512 (FloatTensor.java:21).
        f32_activation: -Dllama.quantizeActivation=false (Q8_0 matrices x f32 activation).  Q4_0 / Q8_0-f32act with 512 raise
        UnsupportedSpecies at the first matmul, as the reference throws."""
        assert vector_bits in (0, 128, 256, 512)
        self.vector_bits = vector_bits
        self.f32_activation = f32_activation
        self.c = cfg
        self.t = tensors          # name -> (raw uint8 ndarray, ggml_type)
        self.cr, self.ci = rope
        c = cfg
        self.q_dim = c["n_heads"] * c["head_size"]
        self.kv_dim = c["n_kv_heads"] * c["head_size"]
        self.kc = np.zeros((c["n_layers"], c["ctx"], self.kv_dim), F32)
        self.vc = np.zeros((c["n_layers"], c["ctx"], self.kv_dim), F32)

    def _mm(self, name, x, d0, d1):
        raw, ty = self.t[name]
        if self.vector_bits and (ty in (GGML_F16, GGML_Q4_0) or (ty == GGML_Q8_0 and self.f32_activation)):
            return matmul_vec(raw, ty, x, d0, d1, self.vector_bits)
        return matmul(raw, ty, x, d0, d1)

    def _f32(self, name, n):
        raw, ty = self.t[name]
        return dequant(raw, ty, n)

    def _mm_rows(self, name, row0, x, d0, d1):
        """InferenceCore.matmulExpert :430-432: rows [row0, row0 + d0) of a stacked [E x d0 x d1] tensor, the same per-row dot."""
        raw, ty = self.t[name]
        bs, ts = BLOCK[ty]
        rb = d1 // bs * ts
        sub = raw.view(np.uint8).reshape(-1)[row0 * rb:(row0 + d0) * rb]
        if self.vector_bits and (ty in (GGML_F16, GGML_Q4_0) or (ty == GGML_Q8_0 and self.f32_activation)):
            return matmul_vec(sub, ty, x, d0, d1, self.vector_bits)
        return matmul(sub, ty, x, d0, d1)

    def _moe_ffn(self, p, x, xb):
        """The MoE block of forwardJavaQwen2MoE (InferenceCore.java:363-415); xb = rmsnorm(x) on entry, returns the new x."""
        c = self.c
        dim, E, topk, mh, sh = c["dim"], c["n_experts"], c["n_experts_used"], c["moe_hidden"], c["hidden"]
        probs = softmax(self._mm(p + "ffn_gate_inp.weight", xb, E, dim))        # :373-374, over all experts
        sel, wts = [], []
        for _ in range(topk):                                                     # :376-390: strict >, first index wins, no renormalisation
            idx = int(np.argmax(probs))
            sel.append(idx)
            wts.append(F32(probs[idx]))
            probs[idx] = -np.inf
        silu = lambda h: (h / (1.0 + np.exp(-h.astype(np.float64))).astype(F32)).astype(F32)
        for e, w in zip(sel, wts):                                                # :392-402
            hb = self._mm_rows(p + "ffn_gate_exps.weight", e * mh, xb, mh, dim)
            hb2 = self._mm_rows(p + "ffn_up_exps.weight", e * mh, xb, mh, dim)
            y = self._mm_rows(p + "ffn_down_exps.weight", e * dim, (silu(hb) * hb2).astype(F32), dim, mh)
            x = ((w * y).astype(F32) + x).astype(F32)                             # saxpyInPlace: a * that + this, two roundings
        hb = self._mm(p + "ffn_gate_shexp.weight", xb, sh, dim)                   # :405-410
        hb2 = self._mm(p + "ffn_up_shexp.weight", xb, sh, dim)
        y = self._mm(p + "ffn_down_shexp.weight", (silu(hb) * hb2).astype(F32), dim, sh)
        gate = seq_sum(self._f32(p + "ffn_gate_inp_shexp.weight", dim) * xb)      # :413 FP32FloatTensor.dot = scalarDot
        sw = F32(1.0) / (F32(1.0) + F32(np.exp(-np.float64(gate))))               # :414
        self.moe_sel, self.moe_w, self.moe_shared_w = sel, wts, F32(sw)
        return ((F32(sw) * y).astype(F32) + x).astype(F32)

    def forward(self, token: int, pos: int, want_logits: bool = True, layer_x: list | None = None):
        c = self.c
        dim, hs, kvd, qd, hid = c["dim"], c["head_size"], self.kv_dim, self.q_dim, c["hidden"]
        H, KVH = c["n_heads"], c["n_kv_heads"]
        kvmul = H // KVH
        eps = c["rms_eps"]
        emb_raw, emb_ty = self.t["token_embd.weight"]
        bs, ts = BLOCK[emb_ty]
        row = emb_raw.view(np.uint8).reshape(-1)[token * dim // bs * ts: (token + 1) * dim // bs * ts]
        x = dequant(row, emb_ty, dim)
        granite = c["arch"] == 3                       # forwardGranite (InferenceCore.java:814-924): llama graph + four scalars
        if granite:
            x = (x * F32(c["embedding_scale"])).astype(F32)
        half = hs // 2
        fcr = self.cr[pos * half:(pos + 1) * half]
        fci = self.ci[pos * half:(pos + 1) * half]
        for l in range(c["n_layers"]):
            p = f"blk.{l}."
            xb = rmsnorm(x, self._f32(p + "attn_norm.weight", dim), eps)
            q = self._mm(p + "attn_q.weight", xb, qd, dim)
            k = self._mm(p + "attn_k.weight", xb, kvd, dim)
            v = self._mm(p + "attn_v.weight", xb, kvd, dim)
            if c["arch"] in (2, 5):   # qwen2 (and qwen2moe :289-291): q/k/v bias, InferenceCore.java:456-459
                q = q + self._f32(p + "attn_q.bias", qd)
                k = k + self._f32(p + "attn_k.bias", kvd)
                v = v + self._f32(p + "attn_v.bias", kvd)
            if c["arch"] in (0, 3):   # InferenceCore.java:75-87, adjacent pairs
                def rot(vec):
                    vv = vec.reshape(-1, half, 2)
                    v0, v1 = vv[:, :, 0], vv[:, :, 1]
                    out = np.empty_like(vv)
                    out[:, :, 0] = v0 * fcr - v1 * fci
                    out[:, :, 1] = v0 * fci + v1 * fcr
                    return out.reshape(-1)
            else:                # InferenceCore.java:594-619, per-head norm + NeoX pairs
                if c["arch"] == 1:
                    qn = self._f32(p + "attn_q_norm.weight", hs)
                    kn = self._f32(p + "attn_k_norm.weight", hs)
                    q = np.concatenate([rmsnorm(q[h * hs:(h + 1) * hs], qn, eps) for h in range(H)])
                    k = np.concatenate([rmsnorm(k[h * hs:(h + 1) * hs], kn, eps) for h in range(KVH)])

                def rot(vec):
                    vv = vec.reshape(-1, 2, half)
                    v0, v1 = vv[:, 0, :], vv[:, 1, :]
                    out = np.empty_like(vv)
                    out[:, 0, :] = v0 * fcr - v1 * fci
                    out[:, 1, :] = v0 * fci + v1 * fcr
                    return out.reshape(-1)
            q, k = rot(q.astype(F32)), rot(k.astype(F32))
            self.kc[l, pos] = k
            self.vc[l, pos] = v
            sqrt_hs = F32(np.sqrt(np.float64(hs)))
            xb = np.zeros(qd, F32)
            for h in range(H):   # InferenceCore.java:98-137
                qh = q[h * hs:(h + 1) * hs]
                kk = self.kc[l, :pos + 1, (h // kvmul) * hs:(h // kvmul + 1) * hs]
                score = seq_sum(kk * qh[None, :], axis=1)
                score = (score * F32(c["attention_scale"])).astype(F32) if granite else score / sqrt_hs
                a = softmax(score)
                vv = self.vc[l, :pos + 1, (h // kvmul) * hs:(h // kvmul + 1) * hs]
                xb[h * hs:(h + 1) * hs] = seq_sum(a[:, None] * vv, axis=0)
            ao = self._mm(p + "attn_output.weight", xb, dim, qd)
            x = x + ((ao * F32(c["residual_scale"])).astype(F32) if granite else ao)
            xb = rmsnorm(x, self._f32(p + "ffn_norm.weight", dim), eps)
            if c["arch"] == 5:
                x = self._moe_ffn(p, x, xb)
                if layer_x is not None:
                    layer_x.append(x.copy())
                continue
            hb = self._mm(p + "ffn_gate.weight", xb, hid, dim)
            hb2 = self._mm(p + "ffn_up.weight", xb, hid, dim)
            sig = (1.0 + np.exp(-hb.astype(np.float64))).astype(F32)   # (float)(1.0 + Math.exp(-value))
            hb = (hb / sig) * hb2
            dn = self._mm(p + "ffn_down.weight", hb.astype(F32), dim, hid)
            x = x + ((dn * F32(c["residual_scale"])).astype(F32) if granite else dn)
            if layer_x is not None:
                layer_x.append(x.copy())
        self.x = x
        if not want_logits:
            return None
        x = rmsnorm(x, self._f32("output_norm.weight", dim), eps)
        name = "output.weight" if "output.weight" in self.t else "token_embd.weight"
        lg = self._mm(name, x, c["vocab"], dim)
        return (lg * F32(c["logit_scale"])).astype(F32) if granite else lg


def argmax(v: np.ndarray) -> int:
    """FloatTensor.argmax — first index of the maximum (FloatTensor.java:138-151)."""
    return int(np.argmax(v))
