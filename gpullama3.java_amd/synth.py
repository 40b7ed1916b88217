"""Synthetic random-weight models in the reference's wire format.

There is no network and no .gguf on disk, so pp/tg are measured on random-weight models of the
named architectures.  Values: f32 N(0, 0.02^2) for matrices, 1 + N(0, 0.02^2) for norm weights;
quantised with ggml's reference Q8_0 / Q4_0 rules into exactly the block layout the reference
reads (J/tensor/GGMLType.java:5-21; Q8_0FloatTensor.java:55-63; Q4_0FloatTensor.java:57-71).
Tensor names follow J/model/loader/LlamaModelLoader.java:83-98 and Qwen3ModelLoader.java:96-118;
metadata keys follow LlamaModelLoader.java:47-63 / Qwen3ModelLoader.java:48-74.

Two generators: NumPy (bit-stable Philox stream, used for the committed golden fixtures) and
torch (any device — the GPU box generates the 1B/8B models in HBM in seconds).
"""
from __future__ import annotations

from dataclasses import dataclass, asdict

import numpy as np

from . import gguf
from .gguf import GGML_F32, GGML_F16, GGML_Q4_0, GGML_Q8_0

ARCH_LLAMA, ARCH_QWEN3, ARCH_QWEN2, ARCH_GRANITE, ARCH_PHI3, ARCH_QWEN2MOE = 0, 1, 2, 3, 4, 5
_ARCH_NAME = {ARCH_LLAMA: "llama", ARCH_QWEN3: "qwen3", ARCH_QWEN2: "qwen2", ARCH_GRANITE: "granite", ARCH_PHI3: "phi3",
              ARCH_QWEN2MOE: "qwen2moe"}


@dataclass
class ModelConfig:
    name: str
    arch: int
    dim: int
    hidden: int
    n_layers: int
    n_heads: int
    n_kv_heads: int
    head_size: int
    vocab: int
    ctx: int
    rms_eps: float
    rope_theta: float
    tied: bool          # wcls shares token_embd (AbstractModelLoader.java:194)
    # Granite only (GraniteLoader.java:55-58); neutral values for every other architecture
    embedding_scale: float = 1.0
    attention_scale: float = 0.0
    residual_scale: float = 1.0
    logit_scale: float = 1.0
    # Devstral 2 only (DevstralModelLoader.java:80-86): (factor, beta_fast, beta_slow, log_multiplier, original_context_length) of
    # the YaRN RoPE table, and the metadata prefix / general.architecture of the file ("mistral3")
    yarn: tuple | None = None
    gguf_arch: str | None = None
    # Qwen2-MoE only (Qwen2MoEModelLoader.java:56-84): expert_count, expert_used_count, the routed experts' hidden size (rows of
    # blk.0.ffn_down_exps.weight's first dimension); ``hidden`` is the shared expert's size (qwen2moe.feed_forward_length)
    n_experts: int = 0
    n_experts_used: int = 0
    moe_hidden: int = 0

    @property
    def q_dim(self):
        return self.n_heads * self.head_size

    @property
    def kv_dim(self):
        return self.n_kv_heads * self.head_size


CONFIGS = {
    # BASELINE.json configs[1] / configs[2] / configs[4] and two tiny fixture shapes
    "llama-3.2-1b": ModelConfig("Llama-3.2-1B-random", ARCH_LLAMA, 2048, 8192, 16, 32, 8, 64, 128256, 648, 1e-5, 500000.0, True),
    "llama-3-8b": ModelConfig("Llama-3-8B-random", ARCH_LLAMA, 4096, 14336, 32, 32, 8, 128, 128256, 648, 1e-5, 500000.0, False),
    "qwen3-4b": ModelConfig("Qwen3-4B-random", ARCH_QWEN3, 2560, 9728, 36, 32, 8, 128, 151936, 648, 1e-6, 1000000.0, True),
    "tiny-llama": ModelConfig("tiny-llama-random", ARCH_LLAMA, 256, 512, 2, 8, 2, 32, 512, 64, 1e-5, 500000.0, False),
    "tiny-llama-tied": ModelConfig("tiny-llama-tied-random", ARCH_LLAMA, 256, 768, 3, 8, 4, 32, 640, 48, 1e-5, 10000.0, True),
    "tiny-qwen3": ModelConfig("tiny-qwen3-random", ARCH_QWEN3, 256, 512, 2, 8, 2, 64, 512, 64, 1e-6, 1000000.0, True),
    # wide enough to exercise full 64-block chunks + ragged tails in the HIP matvec (K = 2560, 4096-wide q)
    "mid-qwen3": ModelConfig("mid-qwen3-random", ARCH_QWEN3, 2560, 1536, 2, 32, 8, 128, 2048, 40, 1e-6, 1000000.0, True),
    "mid-llama": ModelConfig("mid-llama-random", ARCH_LLAMA, 2048, 4096, 2, 32, 8, 64, 4096, 160, 1e-5, 500000.0, False),
    # Qwen2 / Qwen2.5 / DeepSeek-R1-Distill-Qwen shape: q/k/v bias, NeoX RoPE, head_size = dim / heads (forwardJavaQwen2)
    "tiny-qwen2": ModelConfig("tiny-qwen2-random", ARCH_QWEN2, 256, 512, 2, 8, 2, 32, 512, 64, 1e-6, 1000000.0, True),
    "mid-qwen2": ModelConfig("mid-qwen2-random", ARCH_QWEN2, 1536, 4480, 2, 12, 2, 128, 2048, 160, 1e-6, 1000000.0, False),
    # multi-head attention (n_heads == n_kv_heads, kvMul = 1) with head_size 128: Llama-2-7B-style head layout
    # Granite 3.x shape: the Llama graph with the four muP scalars (forwardGranite); values of granite-3.3-2b
    "tiny-granite": ModelConfig("tiny-granite-random", ARCH_GRANITE, 256, 512, 2, 8, 2, 32, 512, 64, 1e-5, 10000.0, True,
                                 embedding_scale=12.0, attention_scale=0.015625, residual_scale=0.22, logit_scale=8.0),
    "mid-granite": ModelConfig("mid-granite-random", ARCH_GRANITE, 2048, 4096, 2, 32, 8, 64, 4096, 160, 1e-5, 10000.0, True,
                                embedding_scale=12.0, attention_scale=0.015625, residual_scale=0.22, logit_scale=8.0),
    # Phi-3 shape: fused attn_qkv / gate|up tensors, NeoX RoPE, multi-head attention (Phi-3-mini: 32 / 32 heads)
    "tiny-phi3": ModelConfig("tiny-phi3-random", ARCH_PHI3, 256, 512, 2, 8, 8, 32, 512, 64, 1e-5, 10000.0, False),
    "mid-phi3": ModelConfig("mid-phi3-random", ARCH_PHI3, 1536, 4096, 2, 12, 4, 128, 2048, 160, 1e-5, 10000.0, False),
    # Phi-3-mini / Phi-3.5-mini head layout: head_size = dim / heads = 96 (not a power of two), multi-head attention
    "phi3-hs96": ModelConfig("phi3-hs96-random", ARCH_PHI3, 768, 2048, 2, 8, 8, 96, 1024, 160, 1e-5, 10000.0, False),
    # Devstral 2 shape (forwardJavaDevstral): the Llama graph with head_size != dim / heads (q_dim 512 / 4096 on dim 256 / 2560) and
    # a YaRN table whose ramp (pairs 17 .. 33 of 64 at head_size 128) and mscale are both active
    "tiny-devstral": ModelConfig("tiny-devstral-random", ARCH_LLAMA, 256, 512, 2, 8, 2, 64, 512, 64, 1e-5, 1000000.0, False,
                                  yarn=(8.0, 32.0, 1.0, 1.0, 4096), gguf_arch="mistral3"),
    "mid-devstral": ModelConfig("mid-devstral-random", ARCH_LLAMA, 2560, 4096, 2, 32, 8, 128, 4096, 160, 1e-5, 1000000.0, False,
                                 yarn=(48.0, 32.0, 1.0, 1.0, 8192), gguf_arch="mistral3"),
    # Qwen1.5-MoE shape (forwardJavaQwen2MoE): qwen2 attention + F32 router over n_experts, top-k routed experts, gated shared
    # expert.  mid: 60 experts / top-4 as Qwen1.5-MoE-A2.7B with narrow experts; a2.7b-moe-layer: one layer at the model's own
    # sizes (dim 2048, 16 / 16 heads of 128, experts 1408, shared expert 5632)
    # Qwen1.5-MoE-A2.7B (14.3 B parameters, 2.7 B active per token): 24 layers, 60 routed experts of 1408, top-4, shared expert 5632
    "qwen1.5-moe-a2.7b": ModelConfig("Qwen1.5-MoE-A2.7B-random", ARCH_QWEN2MOE, 2048, 5632, 24, 16, 16, 128, 151936, 648, 1e-6, 1000000.0,
                                      False, n_experts=60, n_experts_used=4, moe_hidden=1408),
    "tiny-qwen2moe": ModelConfig("tiny-qwen2moe-random", ARCH_QWEN2MOE, 256, 512, 2, 8, 2, 32, 512, 64, 1e-6, 1000000.0, False,
                                  n_experts=8, n_experts_used=2, moe_hidden=128),
    "mid-qwen2moe": ModelConfig("mid-qwen2moe-random", ARCH_QWEN2MOE, 2048, 1536, 2, 16, 16, 128, 2048, 160, 1e-6, 1000000.0, False,
                                 n_experts=60, n_experts_used=4, moe_hidden=384),
    "a2.7b-moe-layer": ModelConfig("Qwen1.5-MoE-A2.7B-1layer-random", ARCH_QWEN2MOE, 2048, 5632, 1, 16, 16, 128, 2048, 160, 1e-6,
                                    1000000.0, False, n_experts=60, n_experts_used=4, moe_hidden=1408),
    "mha-llama": ModelConfig("mha-llama-random", ARCH_LLAMA, 1024, 2048, 2, 8, 8, 128, 1024, 160, 1e-5, 10000.0, False),
    # full-size SHAPES of the BASELINE models with few layers / a small vocabulary, so that the CPU oracle finishes in seconds:
    # one Llama-3-8B layer (K = 14336: 112 tile groups, activation quads == 14 * 256 exactly), the 128256-row vocabulary
    # projection on dim 4096, and two Qwen3-4B layers (K = 2560 ragged, head_size 128 != dim / heads, tied wcls)
    "8b-layer": ModelConfig("Llama-3-8B-1layer-random", ARCH_LLAMA, 4096, 14336, 1, 32, 8, 128, 2048, 648, 1e-5, 500000.0, False),
    "8b-vocab": ModelConfig("Llama-3-8B-vocab-random", ARCH_LLAMA, 4096, 1024, 1, 32, 8, 128, 128256, 64, 1e-5, 500000.0, False),
    # BASELINE configs[1] at its own shape: one Llama-3.2-1B layer (hidden 8192: K = 8192 down projection = 64 tile groups on the
    # 8-producer kernel; head_size 64) under the TIED 128256 x 2048 vocabulary projection
    "1b-layer": ModelConfig("Llama-3.2-1B-1layer-random", ARCH_LLAMA, 2048, 8192, 1, 32, 8, 64, 128256, 648, 1e-5, 500000.0, True),
    "qwen3-4b-2l": ModelConfig("Qwen3-4B-2layer-random", ARCH_QWEN3, 2560, 9728, 2, 32, 8, 128, 4096, 64, 1e-6, 1000000.0, True),
    # ragged everything for the > 64-token GEMM: K = 288 / 864 (9 / 27 blocks: the last K stage holds ONE real block), 288 / 480 / 864 rows
    # (not multiples of the 64- / 96- / 128-row tiles), 9 heads on 3 kv heads
    "ragged-llama": ModelConfig("ragged-llama-random", ARCH_LLAMA, 288, 864, 2, 9, 3, 32, 544, 200, 1e-5, 10000.0, False),
}


# ------------------------------------------------------------------ quantisers (NumPy)
def _round_away(x):
    return np.trunc(x + np.copysign(np.float32(0.5), x))


def quantize_q8_0(w: np.ndarray) -> np.ndarray:
    """ggml quantize_row_q8_0_ref: d = amax/127 (stored f16), q = roundf(x / d)."""
    w = np.ascontiguousarray(w, dtype=np.float32).reshape(-1, 32)
    amax = np.max(np.abs(w), axis=1)
    d = (amax / np.float32(127.0)).astype(np.float32)
    with np.errstate(divide="ignore"):
        inv = np.where(d != 0, np.float32(1.0) / d, np.float32(0)).astype(np.float32)
    q = _round_away(w * inv[:, None]).astype(np.int8)
    out = np.empty((w.shape[0], 34), np.uint8)
    out[:, :2] = d.astype(np.float16).view(np.uint8).reshape(-1, 2)
    out[:, 2:] = q.view(np.uint8)
    return out.reshape(-1)


def quantize_q4_0(w: np.ndarray) -> np.ndarray:
    """ggml quantize_row_q4_0_ref: d = (signed max-magnitude)/-8, q = min(15, (int)(x/d + 8.5));
    elem j<16 in the low nibble of byte j, elem j+16 in the high nibble."""
    w = np.ascontiguousarray(w, dtype=np.float32).reshape(-1, 32)
    idx = np.argmax(np.abs(w), axis=1)
    mx = w[np.arange(w.shape[0]), idx]
    d = (mx / np.float32(-8.0)).astype(np.float32)
    with np.errstate(divide="ignore"):
        inv = np.where(d != 0, np.float32(1.0) / d, np.float32(0)).astype(np.float32)
    x = w * inv[:, None] + np.float32(8.5)
    qi = np.minimum(15, x.astype(np.int32)).astype(np.uint8)
    out = np.empty((w.shape[0], 18), np.uint8)
    out[:, :2] = d.astype(np.float16).view(np.uint8).reshape(-1, 2)
    out[:, 2:] = qi[:, :16] | (qi[:, 16:] << 4)
    return out.reshape(-1)


def encode(w: np.ndarray, ggml_type: int) -> np.ndarray:
    if ggml_type == GGML_F32:
        return np.ascontiguousarray(w, "<f4").view(np.uint8).reshape(-1)
    if ggml_type == GGML_F16:
        return np.ascontiguousarray(w, np.float32).astype("<f2").view(np.uint8).reshape(-1)
    if ggml_type == GGML_Q8_0:
        return quantize_q8_0(w)
    if ggml_type == GGML_Q4_0:
        return quantize_q4_0(w)
    raise ValueError(ggml_type)


def rope_table(ctx: int, head_size: int, theta: float):
    """Host-side RoPE table, as the Java host builds it (RoPE.precomputeFreqsCis, ropeScaling=false,
    J/inference/operation/RoPE.java:6-37): freq in double -> f32, pos*freq in f32, cos/sin in double -> f32."""
    i = np.arange(0, head_size, 2, dtype=np.float64)
    freq = (1.0 / np.power(np.float64(theta), i / np.float64(head_size))).astype(np.float32)
    val = (np.arange(ctx, dtype=np.float32)[:, None] * freq[None, :]).astype(np.float32)
    v64 = val.astype(np.float64)
    return (np.ascontiguousarray(np.cos(v64).astype(np.float32).reshape(-1)),
            np.ascontiguousarray(np.sin(v64).astype(np.float32).reshape(-1)))


def rope_table_yarn(ctx: int, head_size: int, theta: float, factor: float, beta_fast: float, beta_slow: float,
                    log_multiplier: float, original_ctx: int):
    """Host-side YaRN table as DevstralModelLoader.precomputeRopeFrequencies builds it (RoPE.precomputeFreqsCisYaRN,
    J/inference/operation/RoPE.java:39-83); NumPy statement with every Java float kept in f32."""
    f = np.float32
    lnb = f(np.log(np.float64(f(theta))))

    def corr_dim(n_rot):
        ratio = f(f(original_ctx) / f(f(f(n_rot) * f(2.0)) * f(np.pi)))
        return f(f(f(head_size) * f(np.log(np.float64(ratio)))) / f(f(2.0) * lnb))

    low, high = corr_dim(beta_fast), corr_dim(beta_slow)
    fscale = f(f(1.0) / f(factor))
    mscale = f(f(1.0) + f(f(f(0.1) * f(log_multiplier)) * f(np.log(np.float64(f(f(1.0) / fscale)))))) if log_multiplier > 0 else f(1.0)
    i = np.arange(0, head_size, 2, dtype=np.float64)
    extrap = (1.0 / np.power(np.float64(theta), i / np.float64(head_size))).astype(f)
    interp = (fscale * extrap).astype(f)
    y = ((np.arange(head_size // 2, dtype=f) - low).astype(f) / max(f(0.001), f(high - low))).astype(f)
    ramp = (f(1.0) - np.minimum(f(1.0), np.maximum(f(0.0), y))).astype(f)
    freq = ((interp * (f(1.0) - ramp)).astype(f) + (extrap * ramp).astype(f)).astype(f)
    v64 = (np.arange(ctx, dtype=f)[:, None] * freq[None, :]).astype(f).astype(np.float64)
    return (np.ascontiguousarray((np.cos(v64).astype(f) * mscale).astype(f).reshape(-1)),
            np.ascontiguousarray((np.sin(v64).astype(f) * mscale).astype(f).reshape(-1)))


def model_rope(cfg: "ModelConfig"):
    """The table the reference's loader for this model builds: YaRN for a Devstral file, the plain one otherwise."""
    if cfg.yarn:
        return rope_table_yarn(cfg.ctx, cfg.head_size, cfg.rope_theta, *cfg.yarn)
    return rope_table(cfg.ctx, cfg.head_size, cfg.rope_theta)


# ------------------------------------------------------------------ tensor list
def tensor_specs(cfg: ModelConfig, wtype: int):
    """(name, rows, cols, ggml_type, kind) in file order; kind: 'mat' | 'norm'."""
    specs = [("token_embd.weight", cfg.vocab, cfg.dim, wtype, "mat")]
    for l in range(cfg.n_layers):
        p = f"blk.{l}."
        if cfg.arch == ARCH_PHI3:       # Phi3ModelLoader.java:111-116: attn_qkv = q | k | v rows, ffn_up = gate | up rows
            specs += [(p + "attn_norm.weight", 1, cfg.dim, GGML_F32, "norm"),
                      (p + "attn_qkv.weight", cfg.q_dim + 2 * cfg.kv_dim, cfg.dim, wtype, "mat"),
                      (p + "attn_output.weight", cfg.dim, cfg.q_dim, wtype, "mat"),
                      (p + "ffn_norm.weight", 1, cfg.dim, GGML_F32, "norm"),
                      (p + "ffn_down.weight", cfg.dim, cfg.hidden, wtype, "mat"),
                      (p + "ffn_up.weight", 2 * cfg.hidden, cfg.dim, wtype, "mat")]
            continue
        specs += [
            (p + "attn_norm.weight", 1, cfg.dim, GGML_F32, "norm"),
            (p + "attn_q.weight", cfg.q_dim, cfg.dim, wtype, "mat"),
            (p + "attn_k.weight", cfg.kv_dim, cfg.dim, wtype, "mat"),
            (p + "attn_v.weight", cfg.kv_dim, cfg.dim, wtype, "mat"),
            (p + "attn_output.weight", cfg.dim, cfg.q_dim, wtype, "mat"),
        ]
        if cfg.arch in (ARCH_QWEN2, ARCH_QWEN2MOE):
            specs += [(p + "attn_q.bias", 1, cfg.q_dim, GGML_F32, "bias"), (p + "attn_k.bias", 1, cfg.kv_dim, GGML_F32, "bias"),
                      (p + "attn_v.bias", 1, cfg.kv_dim, GGML_F32, "bias")]
        if cfg.arch == ARCH_QWEN3:
            specs += [(p + "attn_q_norm.weight", 1, cfg.head_size, GGML_F32, "norm"),
                      (p + "attn_k_norm.weight", 1, cfg.head_size, GGML_F32, "norm")]
        if cfg.arch == ARCH_QWEN2MOE:      # Qwen2MoEModelLoader.java:97-105; router and shared-expert gate are F32 in the published files
            E, mh = cfg.n_experts, cfg.moe_hidden
            specs += [(p + "ffn_norm.weight", 1, cfg.dim, GGML_F32, "norm"),
                      (p + "ffn_gate_inp.weight", E, cfg.dim, GGML_F32, "router"),
                      (p + "ffn_gate_exps.weight", E * mh, cfg.dim, wtype, "mat"),
                      (p + "ffn_up_exps.weight", E * mh, cfg.dim, wtype, "mat"),
                      (p + "ffn_down_exps.weight", E * cfg.dim, mh, wtype, "mat"),
                      (p + "ffn_gate_shexp.weight", cfg.hidden, cfg.dim, wtype, "mat"),
                      (p + "ffn_up_shexp.weight", cfg.hidden, cfg.dim, wtype, "mat"),
                      (p + "ffn_down_shexp.weight", cfg.dim, cfg.hidden, wtype, "mat"),
                      (p + "ffn_gate_inp_shexp.weight", 1, cfg.dim, GGML_F32, "router")]
            continue
        specs += [
            (p + "ffn_norm.weight", 1, cfg.dim, GGML_F32, "norm"),
            (p + "ffn_gate.weight", cfg.hidden, cfg.dim, wtype, "mat"),
            (p + "ffn_down.weight", cfg.dim, cfg.hidden, wtype, "mat"),
            (p + "ffn_up.weight", cfg.hidden, cfg.dim, wtype, "mat"),
        ]
    specs.append(("output_norm.weight", 1, cfg.dim, GGML_F32, "norm"))
    if not cfg.tied:
        specs.append(("output.weight", cfg.vocab, cfg.dim, wtype, "mat"))
    return specs


class SynthModel:
    """cfg + tensors (name -> (raw uint8 ndarray, ggml_type, rows, cols)) + rope tables."""

    def __init__(self, cfg: ModelConfig, wtype: int, tensors: dict):
        self.cfg, self.wtype, self.tensors = cfg, wtype, tensors
        self.rope = model_rope(cfg)

    def tensor_items(self):
        return self.tensors.items()

    def oracle_tensors(self):
        """name -> (raw, type) with the Phi-3 fused tensors presented as row views q | k | v and gate | up (no copy)."""
        out = {}
        c = self.cfg
        for k, v in self.tensors.items():
            raw, ty, rows, cols = v
            if c.arch == ARCH_PHI3 and k.endswith("attn_qkv.weight"):
                rb = raw.size // rows
                pre = k[: -len("attn_qkv.weight")]
                out[pre + "attn_q.weight"] = (raw[: c.q_dim * rb], ty)
                out[pre + "attn_k.weight"] = (raw[c.q_dim * rb:(c.q_dim + c.kv_dim) * rb], ty)
                out[pre + "attn_v.weight"] = (raw[(c.q_dim + c.kv_dim) * rb:], ty)
            elif c.arch == ARCH_PHI3 and k.endswith("ffn_up.weight"):
                rb = raw.size // rows
                pre = k[: -len("ffn_up.weight")]
                out[pre + "ffn_gate.weight"] = (raw[: c.hidden * rb], ty)
                out[pre + "ffn_up.weight"] = (raw[c.hidden * rb:], ty)
            else:
                out[k] = (raw, ty)
        return out

    def oracle_cfg(self):
        c = self.cfg
        return dict(arch=c.arch, dim=c.dim, hidden=c.hidden, n_layers=c.n_layers, n_heads=c.n_heads,
                    n_kv_heads=c.n_kv_heads, head_size=c.head_size, vocab=c.vocab, ctx=c.ctx, rms_eps=c.rms_eps,
                    embedding_scale=c.embedding_scale, attention_scale=c.attention_scale, residual_scale=c.residual_scale,
                    logit_scale=c.logit_scale, n_experts=c.n_experts, n_experts_used=c.n_experts_used, moe_hidden=c.moe_hidden)

    def weight_bytes(self):
        return sum(v[0].nbytes for v in self.tensors.values())

    # ---- GGUF round trip (metadata keys as the reference loaders read them)
    def metadata(self):
        c = self.cfg
        a = c.gguf_arch or _ARCH_NAME[c.arch]
        ftype = {GGML_F32: 0, GGML_F16: 1, GGML_Q4_0: 2, GGML_Q8_0: 7}[self.wtype]
        md = {
            "general.architecture": a, "general.name": c.name, "general.file_type": ftype,
            f"{a}.embedding_length": c.dim, f"{a}.feed_forward_length": c.hidden, f"{a}.block_count": c.n_layers,
            f"{a}.attention.head_count": c.n_heads, f"{a}.attention.head_count_kv": c.n_kv_heads,
            f"{a}.context_length": c.ctx, f"{a}.attention.layer_norm_rms_epsilon": float(c.rms_eps),
            f"{a}.rope.freq_base": float(c.rope_theta), f"{a}.vocab_size": c.vocab,
        }
        if c.arch == ARCH_GRANITE:
            md.update({"granite.embedding_scale": float(c.embedding_scale), "granite.attention.scale": float(c.attention_scale),
                       "granite.residual_scale": float(c.residual_scale), "granite.logit_scale": float(c.logit_scale)})
        if c.arch == ARCH_QWEN2MOE:
            md.update({f"{a}.expert_count": c.n_experts, f"{a}.expert_used_count": c.n_experts_used,
                       f"{a}.expert_feed_forward_length": c.moe_hidden, f"{a}.expert_shared_feed_forward_length": c.hidden})
        if c.yarn:
            md.update({f"{a}.rope.scaling.type": "yarn", f"{a}.rope.scaling.factor": float(c.yarn[0]),
                       f"{a}.rope.scaling.yarn_beta_fast": float(c.yarn[1]), f"{a}.rope.scaling.yarn_beta_slow": float(c.yarn[2]),
                       f"{a}.rope.scaling.yarn_log_multiplier": float(c.yarn[3]),
                       f"{a}.rope.scaling.original_context_length": int(c.yarn[4])})
        if c.arch == ARCH_QWEN3 or c.head_size * c.n_heads != c.dim:
            md[f"{a}.attention.key_length"] = c.head_size
            md[f"{a}.attention.value_length"] = c.head_size
        return md

    def write_gguf(self, path: str):
        ts = []
        for name, (raw, ty, rows, cols) in self.tensors.items():
            dims = [cols] if rows == 1 and ty == GGML_F32 else [cols, rows]
            if "_exps." in name:                  # stacked experts: [n_experts x rows / n_experts x cols], ne = {cols, rows / E, E}
                dims = [cols, rows // self.cfg.n_experts, self.cfg.n_experts]
            ts.append((name, dims, ty, raw))
        gguf.write_gguf(path, self.metadata(), ts)

    @staticmethod
    def from_gguf(path: str, ctx: int | None = None) -> "SynthModel":
        g = gguf.GGUFFile(path)
        md = g.metadata
        a = md["general.architecture"]
        arch = ARCH_LLAMA if a == "mistral3" else {v: k for k, v in _ARCH_NAME.items()}[a]
        yarn = None
        # only the reference's Devstral loader reads rope.scaling.* (DevstralModelLoader.java:80-93); the llama / qwen2 / qwen3 loaders
        # build the plain table whatever the file says
        if a == "mistral3" and md.get(f"{a}.rope.scaling.type") == "yarn":
            yarn = (md[f"{a}.rope.scaling.factor"], md[f"{a}.rope.scaling.yarn_beta_fast"], md[f"{a}.rope.scaling.yarn_beta_slow"],
                    md.get(f"{a}.rope.scaling.yarn_log_multiplier", 0.0), md[f"{a}.rope.scaling.original_context_length"])
            if not (yarn[0] > 0 and yarn[4] > 0):
                raise ValueError("mistral3.rope.scaling: factor and original_context_length must be > 0")
        dim, nh = md[f"{a}.embedding_length"], md[f"{a}.attention.head_count"]
        hs = md.get(f"{a}.attention.key_length", dim // nh)
        cfg = ModelConfig(md["general.name"], arch, dim, md[f"{a}.feed_forward_length"], md[f"{a}.block_count"], nh,
                          md.get(f"{a}.attention.head_count_kv", nh), hs,
                          md.get(f"{a}.vocab_size", g.tensors["token_embd.weight"][0][1]),
                          ctx or md[f"{a}.context_length"], md[f"{a}.attention.layer_norm_rms_epsilon"],
                          md[f"{a}.rope.freq_base"], "output.weight" not in g.tensors,
                          embedding_scale=md.get("granite.embedding_scale", 1.0), attention_scale=md.get("granite.attention.scale", 0.0),
                          residual_scale=md.get("granite.residual_scale", 1.0), logit_scale=md.get("granite.logit_scale", 1.0),
                          yarn=yarn, gguf_arch=a if a == "mistral3" else None,
                          n_experts=md.get(f"{a}.expert_count", 0), n_experts_used=md.get(f"{a}.expert_used_count", 0),
                          moe_hidden=g.tensors["blk.0.ffn_down_exps.weight"][0][0] if "blk.0.ffn_down_exps.weight" in g.tensors else 0)
        tensors = {}
        for name, (dims, ty, raw) in g.tensors.items():
            rows = int(np.prod(dims[1:])) if len(dims) > 1 else 1
            tensors[name] = (raw, ty, rows, dims[0])
        m = SynthModel(cfg, g.tensors["token_embd.weight"][1], tensors)
        m._gguf = g
        return m


def make_numpy(cfg: ModelConfig, wtype: int = GGML_Q8_0, seed: int = 42, sigma: float = 0.02) -> SynthModel:
    """Bit-stable generator (NumPy Philox, streamed per tensor in file order)."""
    rng = np.random.Generator(np.random.Philox(seed))
    tensors = {}
    for name, rows, cols, ty, kind in tensor_specs(cfg, wtype):
        w = rng.standard_normal((rows, cols), dtype=np.float32) * np.float32(sigma)
        if kind == "norm":
            w = w + np.float32(1.0)
        if kind == "router":              # wider router rows: peaked, token-dependent expert choices
            w = w * np.float32(5.0)
        tensors[name] = (encode(w, ty), ty, rows, cols)
    return SynthModel(cfg, wtype, tensors)


# ------------------------------------------------------------------ torch generator (1B / 8B scale)
def _t_quantize_q8_0(w):
    import torch
    w = w.reshape(-1, 32)
    amax = w.abs().amax(dim=1)
    d = amax / 127.0
    inv = torch.where(d != 0, 1.0 / d, torch.zeros_like(d))
    x = w * inv[:, None]
    q = torch.trunc(x + torch.copysign(torch.full_like(x, 0.5), x)).to(torch.int8)
    out = torch.empty((w.shape[0], 34), dtype=torch.uint8, device=w.device)
    out[:, :2] = d.to(torch.float16).view(torch.uint8).reshape(-1, 2)
    out[:, 2:] = q.view(torch.uint8)
    return out.reshape(-1)


def _t_quantize_q4_0(w):
    import torch
    w = w.reshape(-1, 32)
    idx = w.abs().argmax(dim=1, keepdim=True)
    mx = w.gather(1, idx).squeeze(1)
    d = mx / -8.0
    inv = torch.where(d != 0, 1.0 / d, torch.zeros_like(d))
    qi = torch.clamp((w * inv[:, None] + 8.5).to(torch.int32), max=15).to(torch.uint8)
    out = torch.empty((w.shape[0], 18), dtype=torch.uint8, device=w.device)
    out[:, :2] = d.to(torch.float16).view(torch.uint8).reshape(-1, 2)
    out[:, 2:] = qi[:, :16] | (qi[:, 16:] << 4)
    return out.reshape(-1)


def iter_torch(cfg: ModelConfig, wtype: int = GGML_Q8_0, seed: int = 42, sigma: float = 0.02, device="cpu",
               rows_per_chunk: int = 16384):
    """Streams (name, raw host bytes, ggml_type, rows, cols) in file order, generated with torch RNG on
    ``device``.  Raw bytes are host NumPy arrays because the C-ABI takes caller-owned host pointers (the
    Java host's mmap'd GGUF); a streaming consumer never holds more than one tensor."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    for name, rows, cols, ty, kind in tensor_specs(cfg, wtype):
        parts = []
        for r0 in range(0, rows, rows_per_chunk):
            r = min(rows_per_chunk, rows - r0)
            w = torch.randn((r, cols), generator=g, device=device, dtype=torch.float32) * sigma
            if kind == "norm":
                w = w + 1.0
            if kind == "router":
                w = w * 5.0
            if ty == GGML_F32:
                b = w.contiguous().view(torch.uint8).reshape(-1)
            elif ty == GGML_F16:
                b = w.to(torch.float16).contiguous().view(torch.uint8).reshape(-1)
            elif ty == GGML_Q8_0:
                b = _t_quantize_q8_0(w)
            else:
                b = _t_quantize_q4_0(w)
            parts.append(b.cpu().numpy())
        yield name, (np.concatenate(parts) if len(parts) > 1 else parts[0]), ty, rows, cols


def make_torch(cfg: ModelConfig, wtype: int = GGML_Q8_0, seed: int = 42, sigma: float = 0.02, device="cpu") -> SynthModel:
    """Same recipe as make_numpy with torch RNG on ``device`` (the GPU box builds the 8B model in seconds)."""
    tensors = {name: (raw, ty, rows, cols) for name, raw, ty, rows, cols in iter_torch(cfg, wtype, seed, sigma, device)}
    return SynthModel(cfg, wtype, tensors)


class StreamModel:
    """cfg + rope + a one-shot tensor stream: lets a tensor-parallel rank upload an 8B model without ever
    holding it in host memory (HipMasterPlan consumes ``tensor_items()``)."""

    def __init__(self, cfg: ModelConfig, wtype: int, it):
        self.cfg, self.wtype, self._it = cfg, wtype, it
        self.rope = model_rope(cfg)

    def tensor_items(self):
        for name, raw, ty, rows, cols in self._it:
            yield name, (raw, ty, rows, cols)
