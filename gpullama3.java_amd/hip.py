"""ctypes binding of libgpullama_hip.so (include/gpullama3_hip.h).

This is the Python twin of the JDK-21 FFM binding shown in INTEGRATION.md: same symbols, same
argument order.  There is NO fallback: if the shared library is missing or a symbol is absent the
import of the product path fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_DIR = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("GL3_LIB") or os.path.join(_DIR, "libgpullama_hip.so")      # GL3_LIB: an experimental build of the same library

GL3_OK = 0
E_ARG, E_UNSUPPORTED, E_OOM, E_HIP, E_RCCL, E_STATE = -1, -2, -3, -4, -5, -6
ERR_NAMES = {0: "GL3_OK", -1: "GL3_E_ARG", -2: "GL3_E_UNSUPPORTED", -3: "GL3_E_OOM", -4: "GL3_E_HIP", -5: "GL3_E_RCCL",
             -6: "GL3_E_STATE"}
FLAG_NO_GRAPH, FLAG_LAYER_TAPS, FLAG_FORCE_RCCL, FLAG_SCALAR_DOT, FLAG_F32_ACTIVATION = 1, 2, 4, 8, 16
FLAG_VECTOR_512, FLAG_VECTOR_128 = 32, 64      # -Dllama.VectorBitSize (default 256): see include/gpullama3_hip.h
K_NAMES = ["matvec_qkv", "matvec_wo", "matvec_gateup", "matvec_down", "matvec_logits", "attention", "other", "collective"]

T_IDS = {"token_embd.weight": 0, "output_norm.weight": 1, "output.weight": 2, "attn_norm.weight": 3,
         "attn_q.weight": 4, "attn_k.weight": 5, "attn_v.weight": 6, "attn_output.weight": 7,
         "ffn_norm.weight": 8, "ffn_gate.weight": 9, "ffn_down.weight": 10, "ffn_up.weight": 11,
         "attn_q_norm.weight": 12, "attn_k_norm.weight": 13, "attn_q.bias": 14, "attn_k.bias": 15, "attn_v.bias": 16,
         "attn_qkv.weight": 17,
         # qwen2moe (Qwen2MoEModelLoader.java:97-105): router, stacked experts, shared-expert gate; the shared expert's matrices
         # take the dense FFN ids
         "ffn_gate_inp.weight": 19, "ffn_gate_exps.weight": 20, "ffn_up_exps.weight": 21, "ffn_down_exps.weight": 22,
         "ffn_gate_inp_shexp.weight": 23, "ffn_gate_shexp.weight": 9, "ffn_down_shexp.weight": 10, "ffn_up_shexp.weight": 11}
T_W13 = 18          # phi3: blk.L.ffn_up.weight holds gate | up (forwardJavaPhi3)


class ModelDesc(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("arch", C.c_int32), ("dim", C.c_int32), ("hidden", C.c_int32),
                ("n_layers", C.c_int32), ("n_heads", C.c_int32), ("n_kv_heads", C.c_int32), ("head_size", C.c_int32),
                ("vocab", C.c_int32), ("ctx", C.c_int32), ("rms_eps", C.c_float), ("weight_type", C.c_int32),
                ("max_batch", C.c_int32), ("device", C.c_int32), ("tp_rank", C.c_int32), ("tp_size", C.c_int32),
                ("flags", C.c_uint32), ("n_seqs", C.c_int32), ("embedding_scale", C.c_float), ("attention_scale", C.c_float),
                ("residual_scale", C.c_float), ("logit_scale", C.c_float),
                ("n_experts", C.c_int32), ("n_experts_used", C.c_int32), ("moe_hidden", C.c_int32)]


class KernelTimes(C.Structure):
    _fields_ = [("ms", C.c_double * 8), ("launches", C.c_uint32 * 8), ("bytes", C.c_uint64 * 8)]


_SIGS = {  # symbol -> (restype, argtypes): exactly the declarations of include/gpullama3_hip.h
    "gl3_version": (C.c_char_p, []),
    "gl3_create": (C.c_int32, [C.POINTER(ModelDesc), C.POINTER(C.c_void_p)]),
    "gl3_upload_tensor": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_uint64, C.c_int32]),
    "gl3_upload_rope": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "gl3_tp_peer_access": (C.c_int32, [C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]),
    "gl3_tp_p2p_handle": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "gl3_tp_p2p_attach": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "gl3_tp_unique_id": (C.c_int32, [C.c_void_p, C.c_uint64]),
    "gl3_tp_init": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "gl3_local_group_create": (C.c_int32, [C.c_int32, C.POINTER(C.c_void_p)]),
    "gl3_local_group_destroy": (None, [C.c_void_p]),
    "gl3_tp_attach_local": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "gl3_finalize": (C.c_int32, [C.c_void_p]),
    "gl3_forward_decode": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "gl3_forward_decode_sample": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_int32)]),
    "gl3_get_sample_probs": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "gl3_get_topp_counts": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "gl3_tp_fold_mode": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "gl3_tp_pool_stats": (C.c_int32, [C.c_int32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "gl3_tp_pool_trim": (C.c_int32, [C.c_int32]),
    "gl3_pin_host_buffer": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "gl3_unpin_host_buffer": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "gl3_forward_prefill": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    "gl3_forward_prefill_seq": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32]),
    "gl3_forward_decode_batch": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "gl3_get_kv_seq": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "gl3_get_x": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "gl3_get_layer_x": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p]),
    "gl3_get_kv": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "gl3_get_buffer": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_uint64]),
    "gl3_debug_sumsq": (C.c_int32, [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "gl3_reset_kv": (C.c_int32, [C.c_void_p]),
    "gl3_profile_decode": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(KernelTimes)]),
    "gl3_profile_kernel": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "gl3_profile_prefill_kernel": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "gl3_probe_peaks": (C.c_int32, [C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "gl3_get_init_ms": (C.c_int32, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "gl3_destroy": (None, [C.c_void_p]),
    "gl3_last_error": (C.c_char_p, [C.c_void_p]),
    "gl3_gguf_open": (C.c_int32, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "gl3_gguf_close": (None, [C.c_void_p]),
    "gl3_gguf_last_error": (C.c_char_p, [C.c_void_p]),
    "gl3_gguf_tensor_count": (C.c_int32, [C.c_void_p]),
    "gl3_gguf_tensor_info": (C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_int32), C.POINTER(C.c_uint64),
                                         C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    "gl3_gguf_meta_number": (C.c_int32, [C.c_void_p, C.c_char_p, C.POINTER(C.c_double)]),
    "gl3_gguf_meta_string": (C.c_int32, [C.c_void_p, C.c_char_p, C.POINTER(C.c_char_p)]),
    "gl3_gguf_model_desc": (C.c_int32, [C.c_void_p, C.POINTER(ModelDesc), C.POINTER(C.c_float)]),
    "gl3_rope_table": (None, [C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p]),
    "gl3_rope_table_yarn": (None, [C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32,
                                   C.c_void_p, C.c_void_p]),
    "gl3_gguf_yarn_params": (C.c_int32, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                         C.POINTER(C.c_float), C.POINTER(C.c_int32)]),
    "gl3_kquant_to_q8_0": (C.c_int32, [C.c_int32, C.c_void_p, C.c_uint64, C.c_void_p]),
    "gl3_load_gguf": (C.c_int32, [C.c_char_p, C.POINTER(ModelDesc), C.POINTER(C.c_void_p)]),
}

_lib = None


class Gl3Error(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (ERR_NAMES.get(code, "GL3_E_?"), code, msg))
        self.code = code


def lib():
    """Load libgpullama_hip.so once.  torch (if the process uses it) must be imported first so that both
    share one libamdhip64 / librccl (same SONAME, torch bundles its own copy)."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise ImportError("libgpullama_hip.so is not built: run `python __graft_entry__.py` (hipcc --offload-arch=gfx950); "
                              "there is no CPU fallback for the HIP path")
        try:
            import torch  # noqa: F401  (shares the HIP runtime with torch.distributed / device memory plumbing)
        except Exception:
            pass
        L = C.CDLL(SO_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)           # AttributeError if the symbol is not exported
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check_gguf(code: int, g=None):
    if code != 0:
        msg = lib().gl3_gguf_last_error(g)
        raise Gl3Error(code, msg.decode() if msg else "")


def check_exports():
    """Every symbol include/gpullama3_hip.h declares must be exported (no compute is launched)."""
    import re
    hdr = open(os.path.join(os.path.dirname(_DIR), "include", "gpullama3_hip.h")).read()
    declared = set(re.findall(r"GL3_API[^;]*?\b(gl3_\w+)\s*\(", hdr))
    assert declared == set(_SIGS), (declared ^ set(_SIGS))
    L = C.CDLL(SO_PATH)
    for name in declared:
        getattr(L, name)
    return sorted(declared)


def check(code, ctx=None):
    if code != GL3_OK:
        msg = lib().gl3_last_error(ctx)
        raise Gl3Error(code, msg.decode() if msg else "")
