"""HipMasterPlan — host-side mirror of the reference's plan interface over the C-ABI.

Mirrors ``interface TornadoVMMasterPlan`` (J/tornadovm/TornadoVMMasterPlan.java:30-85) and its
BatchPrefillDecode subclass (J/tornadovm/TornadoVMMasterPlanBatchPrefillDecode.java:107-168) with
the reference's method names, so callers read like the Java engines:

    plan = HipMasterPlan.initializeTornadoVMPlan(model, prefill_batch_size=512)   # ctor + copy-in
    plan.tornadoVMForwardBatchPrefill(tokens, start_pos)                          # no logits
    logits = plan.tornadoVMForwardDecode(token, position)                         # f32[vocab]
    plan.freeTornadoExecutionPlan()

Differences that the C-ABI makes explicit: the token id is an argument (the
reference passes the embedding row through State.embeddingX) and logits are returned, not left in
State.wrapLogits.  Unsupported model x quant x mode combinations raise Gl3Error(GL3_E_UNSUPPORTED)
where the reference throws UnsupportedOperationException (ForwardPlanFactory.java:84-86).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import hip


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class HipMasterPlan:
    def __init__(self, model, prefill_batch_size: int = 1, device: int = 0, tp_rank: int = 0, tp_size: int = 1,
                 flags: int = 0, unique_id: bytes | None = None, local_group=None, n_seqs: int = 1, p2p_exchange=None):
        """model: synth.SynthModel-like — cfg, tensors {gguf name: (raw uint8, ggml_type, rows, cols)}, rope (cr, ci).

        Tensor parallel (tp_size > 1), one of: ``p2p_exchange`` — a callable that takes this rank's 64-byte IPC handle and
        returns the handles of ALL ranks in rank order (e.g. torch.distributed.all_gather_object): peer-write all-gathers over
        xGMI, the default transport; ``unique_id`` — RCCL fall-back; ``local_group`` — ranks are threads of this process (tests)."""
        L = hip.lib()
        c = model.cfg
        self.cfg = c
        self._ctx = C.c_void_p()
        d = hip.ModelDesc(C.sizeof(hip.ModelDesc), c.arch, c.dim, c.hidden, c.n_layers, c.n_heads, c.n_kv_heads,
                          c.head_size, c.vocab, c.ctx, c.rms_eps, model.wtype, prefill_batch_size, device, tp_rank,
                          tp_size, flags, n_seqs, getattr(c, "embedding_scale", 1.0), getattr(c, "attention_scale", 0.0),
                          getattr(c, "residual_scale", 1.0), getattr(c, "logit_scale", 1.0),
                          getattr(c, "n_experts", 0), getattr(c, "n_experts_used", 0), getattr(c, "moe_hidden", 0))
        hip.check(L.gl3_create(C.byref(d), C.byref(self._ctx)))
        self.tp_size, self.tp_rank = tp_size, tp_rank
        self.max_batch = prefill_batch_size
        try:
            if local_group is not None:
                hip.check(L.gl3_tp_attach_local(self._ctx, local_group), self._ctx)
            elif p2p_exchange is not None:
                mine = C.create_string_buffer(64)
                hip.check(L.gl3_tp_p2p_handle(self._ctx, mine, 64), self._ctx)
                every = list(p2p_exchange(mine.raw))
                assert len(every) == tp_size and all(len(h) == 64 for h in every), "p2p_exchange must return tp_size 64-byte handles"
                blob = C.create_string_buffer(b"".join(every), 64 * tp_size)
                hip.check(L.gl3_tp_p2p_attach(self._ctx, blob, 64 * tp_size), self._ctx)
            elif tp_size > 1 or flags & hip.FLAG_FORCE_RCCL:
                assert unique_id is not None, "tensor parallel plan needs the RCCL unique id from rank 0"
                buf = C.create_string_buffer(unique_id, len(unique_id))
                hip.check(L.gl3_tp_init(self._ctx, buf, len(unique_id)), self._ctx)
            for name, t in model.tensor_items():
                raw, ty = t[0], t[1]
                if name.startswith("blk."):
                    _, l, rest = name.split(".", 2)
                    tid, layer = hip.T_IDS[rest], int(l)
                    if c.arch == 4 and rest == "ffn_up.weight":        # phi3: the fused gate | up tensor
                        tid = hip.T_W13
                else:
                    tid, layer = hip.T_IDS[name], 0
                raw = np.ascontiguousarray(raw)
                hip.check(L.gl3_upload_tensor(self._ctx, tid, layer, _p(raw), raw.nbytes, ty), self._ctx)
            cr, ci = model.rope
            hip.check(L.gl3_upload_rope(self._ctx, _p(cr), _p(ci), cr.size), self._ctx)
            self.forceCopyInReadOnlyData()
        except Exception:
            self.freeTornadoExecutionPlan()
            raise
        self._arg = C.c_int32()
        self._pin_logits()

    def _pin_logits(self):
        """The logits buffer is reused on every step: page-lock it once so the D2H copy lands in it directly.  Registration pins
        whole pages, so the buffer is an anonymous mapping of its own (page-aligned, padded to whole pages) — never a slice
        of the C heap, whose pages it would share with unrelated allocations."""
        import mmap
        nbytes = (self.cfg.vocab * 4 + 4095) & ~4095
        self._logits_map = mmap.mmap(-1, nbytes)
        self._logits = np.frombuffer(self._logits_map, np.float32, self.cfg.vocab)
        try:
            hip.check(hip.lib().gl3_pin_host_buffer(self._ctx, _p(self._logits), nbytes), self._ctx)
        except hip.Gl3Error:
            pass                                      # not fatal: the plan falls back to its own staging buffer

    @classmethod
    def from_gguf(cls, path: str, prefill_batch_size: int = 1, ctx: int = 0, device: int = 0, flags: int = 0, n_seqs: int = 1):
        """Native loader (gl3_load_gguf): the library mmaps the GGUF file, reads the config keys the reference loaders read
        (LlamaModelLoader.java:47-69, Qwen3ModelLoader.java:48-79), uploads every tensor and builds the RoPE table itself."""
        from . import synth
        L = hip.lib()
        self = cls.__new__(cls)
        g = C.c_void_p()
        hip.check_gguf(L.gl3_gguf_open(path.encode(), C.byref(g)))
        try:
            d = hip.ModelDesc()
            d.ctx = ctx
            theta = C.c_float()
            hip.check_gguf(L.gl3_gguf_model_desc(g, C.byref(d), C.byref(theta)), g)
            name = C.c_char_p()
            nm = name.value.decode() if L.gl3_gguf_meta_string(g, b"general.name", C.byref(name)) == 0 else os.path.basename(path)
            tied = True
            for i in range(L.gl3_gguf_tensor_count(g)):
                tn = C.c_char_p()
                L.gl3_gguf_tensor_info(g, i, C.byref(tn), None, None, None, None)
                if tn.value == b"output.weight":
                    tied = False
            # Devstral 2 ("mistral3"): the native loader uploads the YaRN table, so the host-side config must describe it too — an
            # oracle or KV comparison built from plan.cfg would otherwise compute the plain table (r4 advisor finding)
            ga = C.c_char_p()
            gguf_arch = ga.value.decode() if L.gl3_gguf_meta_string(g, b"general.architecture", C.byref(ga)) == 0 else None
            yf, ybf, ybs, ylm, yoc = C.c_float(), C.c_float(), C.c_float(), C.c_float(), C.c_int32()
            yr = L.gl3_gguf_yarn_params(g, C.byref(yf), C.byref(ybf), C.byref(ybs), C.byref(ylm), C.byref(yoc))
            if yr < 0:
                raise hip.Gl3Error(hip.E_ARG, "mistral3.rope.scaling: factor and original_context_length must be finite and > 0")
            yarn = (yf.value, ybf.value, ybs.value, ylm.value, yoc.value) if yr > 0 else None
        finally:
            L.gl3_gguf_close(g)
        self.cfg = synth.ModelConfig(nm, d.arch, d.dim, d.hidden, d.n_layers, d.n_heads, d.n_kv_heads, d.head_size, d.vocab, d.ctx,
                                     d.rms_eps, float(theta.value), tied, n_experts=d.n_experts, n_experts_used=d.n_experts_used,
                                     moe_hidden=d.moe_hidden, yarn=yarn, gguf_arch=gguf_arch if gguf_arch == "mistral3" else None)
        opts = hip.ModelDesc()
        opts.struct_size = C.sizeof(hip.ModelDesc)
        opts.ctx, opts.max_batch, opts.device, opts.tp_size, opts.flags, opts.n_seqs = ctx, prefill_batch_size, device, 1, flags, n_seqs
        self._ctx = C.c_void_p()
        hip.check_gguf(L.gl3_load_gguf(path.encode(), C.byref(opts), C.byref(self._ctx)))
        self.tp_size, self.tp_rank, self.max_batch = 1, 0, prefill_batch_size
        self._arg = C.c_int32()
        self._pin_logits()
        return self

    # ---- reference-named interface -------------------------------------------------------------
    @classmethod
    def initializeTornadoVMPlan(cls, model, prefill_batch_size: int = 1, **kw) -> "HipMasterPlan":
        """TornadoVMMasterPlan.initializeTornadoVMPlan(state, model) :55-70 — the plan flavour is picked from
        llama.prefillBatchSize: > 1 allocates the batched-prefill (MFMA) buffers."""
        return cls(model, prefill_batch_size=prefill_batch_size, **kw)

    def forceCopyInReadOnlyData(self):
        hip.check(hip.lib().gl3_finalize(self._ctx), self._ctx)

    def tornadoVMForwardDecode(self, token: int, position: int) -> np.ndarray:
        """One decode step; returns the logits (host-visible on return, like state.wrapLogits)."""
        return self.forward_decode(token, position)

    def tornadoVMForwardPrefill(self, token: int, position: int):
        """TornadoVMMasterPlanPrefillDecode.tornadoVMForwardPrefill(position): one token, logits skipped."""
        t = np.array([token], np.int32)
        hip.check(hip.lib().gl3_forward_prefill(self._ctx, _p(t), 1, position), self._ctx)

    def tornadoVMForwardBatchPrefill(self, tokens, start_pos: int):
        """TornadoVMMasterPlanBatchPrefillDecode.tornadoVMForwardBatchPrefill(): one chunk, logits skipped."""
        t = np.ascontiguousarray(tokens, np.int32)
        hip.check(hip.lib().gl3_forward_prefill(self._ctx, _p(t), t.size, start_pos), self._ctx)

    def prefill_seq(self, seq: int, tokens, start_pos: int = 0):
        """Prefill sequence `seq` (its own KV cache) in chunks of max_batch."""
        tokens = list(tokens)
        b = max(1, self.max_batch)
        for off in range(0, len(tokens), b):
            t = np.ascontiguousarray(tokens[off:off + b], np.int32)
            hip.check(hip.lib().gl3_forward_prefill_seq(self._ctx, seq, _p(t), t.size, start_pos + off), self._ctx)

    def forward_decode_batch(self, tokens, seq_ids, positions, want_logits: bool = True):
        """Static batched decode: one step for n independent sequences -> (logits [n][vocab] or None, greedy ids [n])."""
        t = np.ascontiguousarray(tokens, np.int32); s = np.ascontiguousarray(seq_ids, np.int32); p = np.ascontiguousarray(positions, np.int32)
        n = t.size
        logits = np.empty((n, self.cfg.vocab), np.float32) if want_logits else None
        ids = np.empty(n, np.int32)
        hip.check(hip.lib().gl3_forward_decode_batch(self._ctx, _p(t), _p(s), _p(p), n, _p(logits) if want_logits else None, _p(ids)), self._ctx)
        return logits, ids

    def kv_seq(self, seq: int, layer: int, pos: int):
        n = self.cfg.kv_dim // self.tp_size
        k, v = np.empty(n, np.float32), np.empty(n, np.float32)
        hip.check(hip.lib().gl3_get_kv_seq(self._ctx, seq, layer, pos, _p(k), _p(v)), self._ctx)
        return k, v

    def freeTornadoExecutionPlan(self):
        if getattr(self, "_ctx", None) is not None and self._ctx:
            hip.lib().gl3_destroy(self._ctx)
            self._ctx = C.c_void_p()

    # ---- pythonic conveniences -----------------------------------------------------------------
    def forward_decode(self, token: int, position: int, copy: bool = True):
        hip.check(hip.lib().gl3_forward_decode(self._ctx, token, position, _p(self._logits), None), self._ctx)
        return self._logits.copy() if copy else self._logits

    def forward_decode_argmax(self, token: int, position: int) -> int:
        """-Dllama.deviceSample: greedy token id sampled on the device (4 bytes D2H instead of vocab*4)."""
        hip.check(hip.lib().gl3_forward_decode(self._ctx, token, position, None, C.byref(self._arg)), self._ctx)
        return int(self._arg.value)

    def forward_decode_sample(self, token: int, position: int, temperature: float, topp: float, coin: float) -> int:
        """One decode step + Sampler.selectSampler(...).sampleToken(logits) (Sampler.java:76-123); `coin` = rng.nextFloat(1f) from the
        caller's RandomGenerator (javarand.L32X64MixRandom mirrors RandomGeneratorFactory.getDefault())."""
        out = C.c_int32()
        hip.check(hip.lib().gl3_forward_decode_sample(self._ctx, token, position, temperature, topp, coin, C.byref(out)), self._ctx)
        return int(out.value)

    def sample_probs(self) -> np.ndarray:
        out = np.empty(self.cfg.vocab, np.float32)
        hip.check(hip.lib().gl3_get_sample_probs(self._ctx, _p(out)), self._ctx)
        return out

    def prefill(self, tokens, start_pos: int = 0, batch: int | None = None):
        """LlamaBench.prefill (J/bench/LlamaBench.java:258-273): chunks of `batch` (llama-bench -b; default max_batch)."""
        tokens = list(tokens)
        b = max(1, min(batch or self.max_batch, max(1, self.max_batch)))
        for off in range(0, len(tokens), b):
            self.tornadoVMForwardBatchPrefill(tokens[off:off + b], start_pos + off)

    def tp_fold_mode(self):
        """(mode, consumer mask) of the tensor-parallel hand-over of this plan's decode step (gl3_tp_fold_mode)."""
        a, b = C.c_int32(), C.c_int32()
        hip.check(hip.lib().gl3_tp_fold_mode(self._ctx, C.byref(a), C.byref(b)), self._ctx)
        return a.value, b.value

    def topp_counts(self):
        """(top-p draws answered on the device, draws answered by the host heap after a tie at the sampled rank)."""
        a, b = C.c_int64(), C.c_int64()
        hip.check(hip.lib().gl3_get_topp_counts(self._ctx, C.byref(a), C.byref(b)), self._ctx)
        return a.value, b.value

    def x(self):
        out = np.empty(self.cfg.dim, np.float32)
        hip.check(hip.lib().gl3_get_x(self._ctx, _p(out)), self._ctx)
        return out

    def layer_x(self, layer: int):
        out = np.empty(self.cfg.dim, np.float32)
        hip.check(hip.lib().gl3_get_layer_x(self._ctx, layer, _p(out)), self._ctx)
        return out

    def kv(self, layer: int, pos: int):
        n = self.cfg.kv_dim // self.tp_size
        k, v = np.empty(n, np.float32), np.empty(n, np.float32)
        hip.check(hip.lib().gl3_get_kv(self._ctx, layer, pos, _p(k), _p(v)), self._ctx)
        return k, v

    def buffer(self, which: int, n: int):
        out = np.empty(n, np.float32)
        hip.check(hip.lib().gl3_get_buffer(self._ctx, which, _p(out), n), self._ctx)
        return out

    def reset_kv(self):
        hip.check(hip.lib().gl3_reset_kv(self._ctx), self._ctx)

    def profile_decode(self, token: int, position: int) -> dict:
        kt = hip.KernelTimes()
        hip.check(hip.lib().gl3_profile_decode(self._ctx, token, position, C.byref(kt)), self._ctx)
        return {n: dict(ms=kt.ms[i], launches=kt.launches[i], bytes=kt.bytes[i]) for i, n in enumerate(hip.K_NAMES)}

    def profile_kernel(self, klass: str, iters: int = 10) -> dict:
        """One kernel class, back-to-back over every layer's weights, one HIP event pair (see gl3_profile_kernel)."""
        us, nb = C.c_double(), C.c_uint64()
        hip.check(hip.lib().gl3_profile_kernel(self._ctx, hip.K_NAMES.index(klass), iters, C.byref(us), C.byref(nb)), self._ctx)
        return dict(avg_us=us.value, bytes_per_launch=nb.value, gbs=nb.value / us.value / 1e3)

    def profile_prefill_kernel(self, klass: str, n_tokens: int, iters: int = 3) -> dict:
        """One batched-prefill GEMM class at n_tokens tokens (see gl3_profile_prefill_kernel): device time and int8 TOP/s."""
        us, ops = C.c_double(), C.c_uint64()
        hip.check(hip.lib().gl3_profile_prefill_kernel(self._ctx, hip.K_NAMES.index(klass), n_tokens, iters, C.byref(us), C.byref(ops)), self._ctx)
        return dict(avg_us=us.value, int8_ops_per_launch=ops.value, tops=ops.value / us.value / 1e6)

    def init_ms(self):
        a, b = C.c_double(), C.c_double()
        hip.check(hip.lib().gl3_get_init_ms(self._ctx, C.byref(a), C.byref(b)), self._ctx)
        return dict(plan_creation_ms=a.value, weights_copy_in_ms=b.value)

    def __del__(self):
        try:
            self.freeTornadoExecutionPlan()
        except Exception:
            pass


def make_local_group(n: int):
    """Test transport: n plans in one process on one GPU (one host thread per rank)."""
    g = C.c_void_p()
    hip.check(hip.lib().gl3_local_group_create(n, C.byref(g)))
    return g


def make_unique_id() -> bytes:
    buf = C.create_string_buffer(128)
    hip.check(hip.lib().gl3_tp_unique_id(buf, 128))
    return buf.raw
