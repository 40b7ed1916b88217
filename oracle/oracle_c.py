"""ctypes wrapper over oracle/libgl3_oracle.so (the C restatement; see gl3_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg — never by gpullama3.java_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("GL3_ORACLE_LIB") or os.path.join(_DIR, "libgl3_oracle.so")      # GL3_ORACLE_LIB: the sanitizer build (scripts/sanitize.sh)

T_IDS = {"token_embd.weight": 0, "output_norm.weight": 1, "output.weight": 2, "attn_norm.weight": 3,
         "attn_q.weight": 4, "attn_k.weight": 5, "attn_v.weight": 6, "attn_output.weight": 7,
         "ffn_norm.weight": 8, "ffn_gate.weight": 9, "ffn_down.weight": 10, "ffn_up.weight": 11,
         "attn_q_norm.weight": 12, "attn_k_norm.weight": 13, "attn_q.bias": 14, "attn_k.bias": 15, "attn_v.bias": 16,
         # qwen2moe (Qwen2MoEModelLoader.java:97-105); the shared expert's matrices take the dense FFN slots
         "ffn_gate_inp.weight": 19, "ffn_gate_exps.weight": 20, "ffn_up_exps.weight": 21, "ffn_down_exps.weight": 22,
         "ffn_gate_inp_shexp.weight": 23, "ffn_gate_shexp.weight": 9, "ffn_down_shexp.weight": 10, "ffn_up_shexp.weight": 11}


class OrcConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("arch", "dim", "hidden", "n_layers", "n_heads", "n_kv_heads",
                                          "head_size", "vocab", "ctx")] + [("rms_eps", C.c_float), ("embedding_scale", C.c_float),
                                                                           ("attention_scale", C.c_float), ("residual_scale", C.c_float),
                                                                           ("logit_scale", C.c_float)] + \
               [(n, C.c_int32) for n in ("n_experts", "n_experts_used", "moe_hidden")]


def build(force: bool = False):
    src = os.path.join(_DIR, "gl3_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _DIR, "-B"])
    return _SO


_lib = None
_MAX_THREADS = [0]      # the OpenMP pool size the process started with (filled by lib())


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.orc_create.restype = C.c_void_p
        _MAX_THREADS[0] = L.orc_num_threads()
        L.orc_create.argtypes = [C.POINTER(OrcConfig)]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_get_moe_routing.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_get_moe_routing.restype = None
        L.orc_set_tensor.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_set_rope.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_forward.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_prefill.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.orc_get_x.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_get_kv.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_argmax.argtypes = [C.c_void_p, C.c_int]
        L.orc_rope_table.argtypes = [C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
        L.orc_rope_table_yarn.argtypes = [C.c_int, C.c_int, C.c_double, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int,
                                          C.c_void_p, C.c_void_p]
        L.orc_rope_table_yarn.restype = None
        L.orc_f16_to_f32.restype = C.c_float
        L.orc_f16_to_f32.argtypes = [C.c_uint16]
        L.orc_f32_to_f16.restype = C.c_uint16
        L.orc_f32_to_f16.argtypes = [C.c_float]
        L.orc_quantize_act.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_dot_q8.restype = C.c_float
        L.orc_dot_q8.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_get_float.restype = C.c_float
        L.orc_get_float.argtypes = [C.c_void_p, C.c_int, C.c_long]
        L.orc_rmsnorm.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float]
        L.orc_softmax.argtypes = [C.c_void_p, C.c_int]
        L.orc_sample.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p]
        L.orc_set_vector_bits.argtypes = [C.c_void_p, C.c_int]
        L.orc_set_f32_activation.argtypes = [C.c_void_p, C.c_int]
        L.orc_dot_v256.restype = C.c_float
        L.orc_dot_v256.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orc_dot_vec.restype = C.c_float
        L.orc_dot_vec.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.orc_species_error.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class UnsupportedSpecies(Exception):
    """The reference's UnsupportedOperationException(F_SPECIES.toString()): Q8_0FloatTensor.java:165-167, Q4_0FloatTensor.java:118-120 —
    the Q8_0 (f32 activation) / Q4_0 vector dots exist for 128- and 256-bit species only."""


class COracle:
    """Same call surface as the HIP plan: forward(token, pos) -> logits, prefill(tokens, start)."""

    def __init__(self, model, vector_bits: int = 0, f32_activation: bool = False):
        """model: gpullama3.java_amd synth.SynthModel-like (cfg, tensors name->(raw, type, ...), rope).
        vector_bits: 0 = scalar dots everywhere (-Dllama.VectorBitSize=0); 128 / 256 / 512 = the Vector-API dots of that species for
        F16 / Q4_0 matrices (VectorShape.preferredShape(): 256 on an AVX2 host, 512 on AVX-512 such as the GPU box's EPYC 9575F).
        f32_activation: -Dllama.quantizeActivation=false — Q8_0 matrices take the f32 activation (with vector_bits != 0:
        Q8_0FloatTensor.vectorDot).  Q4_0 / Q8_0-f32act with 512 raise UnsupportedSpecies from forward / prefill, as the reference throws."""
        L = lib()
        c = model.cfg
        self.cfg = c
        oc = OrcConfig(c.arch, c.dim, c.hidden, c.n_layers, c.n_heads, c.n_kv_heads, c.head_size, c.vocab, c.ctx, c.rms_eps,
                       getattr(c, "embedding_scale", 1.0), getattr(c, "attention_scale", 0.0), getattr(c, "residual_scale", 1.0),
                       getattr(c, "logit_scale", 1.0), getattr(c, "n_experts", 0), getattr(c, "n_experts_used", 0), getattr(c, "moe_hidden", 0))
        self._h = L.orc_create(C.byref(oc))
        assert L.orc_set_vector_bits(self._h, vector_bits) == 0
        L.orc_set_f32_activation(self._h, 1 if f32_activation else 0)
        tensors = model.oracle_tensors() if hasattr(model, "oracle_tensors") else model.tensors      # phi3: fused tensors as row views
        self._keep = [model, tensors]
        for name, t in tensors.items():
            raw, ty = t[0], t[1]
            raw = np.ascontiguousarray(raw)
            self._keep.append(raw)
            if name.startswith("blk."):
                _, l, rest = name.split(".", 2)
                L.orc_set_tensor(self._h, T_IDS[rest], int(l), _p(raw), ty)
            else:
                L.orc_set_tensor(self._h, T_IDS[name], 0, _p(raw), ty)
        if "output.weight" not in tensors:      # tied: wcls = token_embd
            raw, ty = tensors["token_embd.weight"][:2]
            L.orc_set_tensor(self._h, T_IDS["output.weight"], 0, _p(raw), ty)
        self._cr, self._ci = model.rope
        L.orc_set_rope(self._h, _p(self._cr), _p(self._ci))

    def _pool(self):
        """OpenMP threads for this model's forwards: every logical CPU for the full-size shapes, 8 for the small test models — a
        256-thread region per matmul of a 256 x 512 matrix spends its time in the barrier (measured on the GPU box: 48 s against
        < 1 s for a 24-step loop of tiny-llama).  Results do not depend on the count: rows and heads are independent."""
        c = self.cfg
        small = c.dim * max(c.hidden, c.vocab) < (1 << 24)
        lib().orc_set_num_threads(8 if small else _MAX_THREADS[0])

    def forward(self, token: int, pos: int, layer_x: bool = False):
        c = self.cfg
        self._pool()
        logits = np.empty(c.vocab, np.float32)
        lx = np.empty((c.n_layers, c.dim), np.float32) if layer_x else None
        lib().orc_forward(self._h, token, pos, _p(logits), _p(lx) if layer_x else None)
        self._check_species()
        return (logits, lx) if layer_x else logits

    def _check_species(self):
        if lib().orc_species_error(self._h):
            raise UnsupportedSpecies("Q8_0 (f32 activation) / Q4_0 vector dots: 128- and 256-bit species only")

    def prefill(self, tokens, start_pos: int):
        self._pool()
        t = np.ascontiguousarray(tokens, np.int32)
        lib().orc_prefill(self._h, _p(t), len(t), start_pos)
        self._check_species()

    def moe_routing(self):
        """(expert ids, routing weights, shared-expert gate) of the last layer of the last step."""
        k = self.cfg.n_experts_used
        sel, w, sw = np.empty(k, np.int32), np.empty(k, np.float32), C.c_float()
        lib().orc_get_moe_routing(self._h, _p(sel), _p(w), C.byref(sw))
        return sel, w, np.float32(sw.value)

    def x(self):
        out = np.empty(self.cfg.dim, np.float32)
        lib().orc_get_x(self._h, _p(out))
        return out

    def kv(self, layer: int, pos: int):
        kvd = self.cfg.n_kv_heads * self.cfg.head_size
        k, v = np.empty(kvd, np.float32), np.empty(kvd, np.float32)
        lib().orc_get_kv(self._h, layer, pos, _p(k), _p(v))
        return k, v

    def close(self):
        if self._h:
            lib().orc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def argmax(v: np.ndarray) -> int:
    v = np.ascontiguousarray(v, np.float32)
    return lib().orc_argmax(_p(v), v.size)


def sample(logits: np.ndarray, temperature: float, topp: float, coin: float, want_probs: bool = False):
    """Sampler.selectSampler(vocab, temperature, topp, .).sampleToken(logits) with rng.nextFloat(1f) = coin."""
    v = np.ascontiguousarray(logits, np.float32)
    probs = np.empty_like(v) if want_probs else None
    tok = lib().orc_sample(_p(v), v.size, temperature, topp, coin, _p(probs) if want_probs else None)
    return (tok, probs) if want_probs else tok
