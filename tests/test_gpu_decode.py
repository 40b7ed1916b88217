"""Parity of the HIP decode path (through the C-ABI) against the CPU oracle and the golden fixtures.

Bar: BIT-EXACT.  BASELINE.json's north star asks for logits within 1e-3 relative and identical greedy ids;
because the reference re-quantises activations to int8 before every matmul, a 1-ulp difference in any f32
reduction grows to ~1e-2 in the logits (measured), so the HIP kernels evaluate every reduction in
the reference's order and the tests assert exact equality of logits, per-layer activations and the KV cache
(np.array_equal on f32), which implies both north-star conditions.
"""
import os

import numpy as np
import pytest

import __graft_entry__ as ge

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))


@pytest.fixture(scope="module")
def planmod():
    from importlib import import_module
    ge.load_package()
    return import_module(ge.PKG_NAME + ".plan"), import_module(ge.PKG_NAME + ".hip")


@pytest.mark.parametrize("fx,cfg,seed,wtype,scalar", [("tiny_llama_q8_0", "tiny-llama", 7, 8, False), ("tiny_qwen3_q8_0", "tiny-qwen3", 5, 8, False),
                                                      ("tiny_llama_f16", "tiny-llama", 7, 1, True), ("tiny_llama_tied_q4_0", "tiny-llama-tied", 11, 2, True),
                                                      ("tiny_qwen2_q8_0", "tiny-qwen2", 13, 8, False), ("tiny_granite_q8_0", "tiny-granite", 19, 8, False), ("tiny_phi3_q8_0", "tiny-phi3", 23, 8, False),
                                                      ("tiny_devstral_q8_0", "tiny-devstral", 29, 8, False),
                                                      ("tiny_llama_f16_v256", "tiny-llama", 7, 1, False), ("tiny_llama_tied_q4_0_v256", "tiny-llama-tied", 11, 2, False),
                                                      ("tiny_llama_q8_0_f32act_v256", "tiny-llama", 7, 8, "f32act")])
def test_decode_matches_golden_fixture(pkg, planmod, fx, cfg, seed, wtype, scalar):
    """F16 / Q4_0: the *_v256 fixtures are the reference's default Vector-API dot order (the plan's default), the others its
    scalar order (GL3_FLAG_SCALAR_DOT)."""
    plan_mod, hip = planmod
    g = np.load(os.path.join(GOLD, fx + ".npz"))
    m = pkg.synth.make_numpy(pkg.synth.CONFIGS[cfg], wtype=wtype, seed=seed)
    mode = hip.FLAG_F32_ACTIVATION if scalar == "f32act" else hip.FLAG_SCALAR_DOT if scalar else 0      # f32act: -Dllama.quantizeActivation=false
    plan = plan_mod.HipMasterPlan(m, flags=hip.FLAG_LAYER_TAPS | mode)
    toks = g["tokens"]
    n_prompt = len(g["prompt"])
    for pos in range(g["logits"].shape[0]):
        lg = plan.tornadoVMForwardDecode(int(toks[pos]), pos)
        assert np.array_equal(lg, g["logits"][pos]), (pos, rel(lg, g["logits"][pos]))
        if pos >= n_prompt - 1:
            assert int(np.argmax(lg)) == toks[pos + 1], pos          # greedy ids identical
    for l in range(m.cfg.n_layers):
        assert np.array_equal(plan.layer_x(l), g["last_layer_x"][l])
        k, v = plan.kv(l, g["logits"].shape[0] - 1)
        assert np.array_equal(k, g["k_last"][l]) and np.array_equal(v, g["v_last"][l])
    plan.freeTornadoExecutionPlan()


@pytest.mark.parametrize("cfg", ["mid-llama", "mid-qwen3", "tiny-llama-tied", "tiny-qwen2", "mid-qwen2", "mha-llama", "tiny-granite", "mid-granite", "tiny-phi3", "mid-phi3",
                                 "phi3-hs96", "mid-devstral"])      # head_size 96 = Phi-3-mini's dim / heads (not a power of two)
def test_decode_matches_c_oracle_live(pkg, orc, planmod, cfg):
    """Shapes with full 64-block chunks, ragged chunk tails (K = 2560), head sizes 32/64/128, tied wcls, and multi-head
    attention (kvMul = 1 with head_size 128: the KV write must cover head_size > 64 * kvMul)."""
    plan_mod, hip = planmod
    m = pkg.synth.make_numpy(pkg.synth.CONFIGS[cfg], seed=21)
    plan = plan_mod.HipMasterPlan(m, flags=hip.FLAG_LAYER_TAPS)
    o = orc.COracle(m)
    toks = pkg.javarand.bench_tokens(m.cfg.vocab, 20)
    for pos, t in enumerate(toks):
        ref, lx = o.forward(t, pos, layer_x=True)
        got = plan.tornadoVMForwardDecode(t, pos)
        assert np.array_equal(got, ref), (pos, rel(got, ref))
        for l in range(m.cfg.n_layers):
            assert np.array_equal(plan.layer_x(l), lx[l])
        assert plan.forward_decode_argmax(t, pos) == orc.argmax(ref)     # device argmax = first index of the max
    plan.freeTornadoExecutionPlan()


@pytest.mark.parametrize("cfg,wtype,scalar", [("mid-llama", 1, True), ("mid-qwen3", 2, True), ("tiny-llama-tied", 1, True), ("mid-llama", 2, True),
                                              ("mid-llama", 1, False), ("mid-qwen3", 2, False), ("tiny-llama-tied", 1, False), ("mid-llama", 2, False),
                                              ("mid-qwen3", 1, False), ("mha-llama", 2, False), ("mid-granite", 1, False), ("mid-granite", 2, True), ("mid-phi3", 1, False), ("mid-phi3", 2, False)])
def test_f16_and_q4_0_decode_match_c_oracle_live(pkg, orc, planmod, cfg, wtype, scalar):
    """F16 and Q4_0 weights (no activation quantisation) in both dot orders of the reference:
    the default Vector-API order with a 256-bit species (FP16FloatTensor.vectorDot / Q4_0FloatTensor.vectorDot: 8 fused
    accumulator lanes per row, matvec_vl_kernel) and the scalar order (-Dllama.VectorBitSize=0: one K-long chain per row,
    matvec_rl_kernel, GL3_FLAG_SCALAR_DOT).  Logits, per-layer x and device argmax bit-identical to the oracle in the
    same mode; prefill of these types runs token by token and must leave the same KV cache."""
    plan_mod, hip = planmod
    m = pkg.synth.make_numpy(pkg.synth.CONFIGS[cfg], wtype=wtype, seed=17)
    plan = plan_mod.HipMasterPlan(m, prefill_batch_size=16, flags=hip.FLAG_LAYER_TAPS | (hip.FLAG_SCALAR_DOT if scalar else 0))
    o = orc.COracle(m, vector_bits=0 if scalar else 256)
    toks = pkg.javarand.bench_tokens(m.cfg.vocab, 14)
    plan.prefill(toks[:6], 0)
    o.prefill(toks[:6], 0)
    for pos in range(6, 14):
        ref, lx = o.forward(toks[pos], pos, layer_x=True)
        got = plan.tornadoVMForwardDecode(toks[pos], pos)
        assert np.array_equal(got, ref), (pos, rel(got, ref))
        for l in range(m.cfg.n_layers):
            assert np.array_equal(plan.layer_x(l), lx[l])
        assert plan.forward_decode_argmax(toks[pos], pos) == orc.argmax(ref)
    for l in range(m.cfg.n_layers):
        k, v = plan.kv(l, 3)
        ko, vo = o.kv(l, 3)
        assert np.array_equal(k, ko) and np.array_equal(v, vo)
    plan.freeTornadoExecutionPlan()


@pytest.mark.parametrize("cfg", ["mid-llama", "mid-qwen3", "tiny-llama-tied", "mid-qwen2", "mid-phi3", "1b-layer"])
def test_f16_on_a_512_bit_species_matches_c_oracle_live(pkg, orc, planmod, cfg):
    """FloatTensor.java:21 takes VectorShape.preferredShape(): on the GPU box's own EPYC 9575F (AVX-512) the reference's F16 dot keeps 16
    accumulator lanes (FP16FloatTensor.vectorDot :63-110 is species-generic) — BASELINE configs[0]'s arithmetic on this node.
    GL3_FLAG_VECTOR_512: logits, per-layer x, device argmax and the KV cache after a (token-by-token) prefill chunk equal the oracle's
    vector_bits = 512 mode bit for bit, and differ from the 256-bit order."""
    plan_mod, hip = planmod
    m = pkg.synth.make_numpy(pkg.synth.CONFIGS[cfg], wtype=1, seed=19)
    plan = plan_mod.HipMasterPlan(m, prefill_batch_size=16, flags=hip.FLAG_LAYER_TAPS | hip.FLAG_VECTOR_512)
    o, o256 = orc.COracle(m, vector_bits=512), orc.COracle(m, vector_bits=256)
    toks = pkg.javarand.bench_tokens(m.cfg.vocab, 12)
    plan.prefill(toks[:6], 0)
    o.prefill(toks[:6], 0); o256.prefill(toks[:6], 0)
    differs = False
    for pos in range(6, 12):
        ref, lx = o.forward(toks[pos], pos, layer_x=True)
        got = plan.tornadoVMForwardDecode(toks[pos], pos)
        assert np.array_equal(got, ref), (pos, rel(got, ref))
        differs |= not np.array_equal(got, o256.forward(toks[pos], pos))
        for l in range(m.cfg.n_layers):
            assert np.array_equal(plan.layer_x(l), lx[l])
        assert plan.forward_decode_argmax(toks[pos], pos) == orc.argmax(ref)
    assert differs
    for l in range(m.cfg.n_layers):
        k, v = plan.kv(l, 3)
        ko, vo = o.kv(l, 3)
        assert np.array_equal(k, ko) and np.array_equal(v, vo)
    plan.freeTornadoExecutionPlan()


def test_vector_species_the_reference_cannot_run_are_refused(pkg, planmod):
    """Q4_0FloatTensor.vectorDot / Q8_0FloatTensor.vectorDot throw UnsupportedOperationException on a 512-bit species (:118-120, :165-167):
    gl3_create answers GL3_E_UNSUPPORTED (-2) for such a plan; the 128-bit species exists in the oracles only and is refused for every
    species-dependent type; the headline Q8_0 int8 path does not depend on the species and accepts the flag."""
    plan_mod, hip = planmod
    for wt, flags in [(2, hip.FLAG_VECTOR_512), (8, hip.FLAG_VECTOR_512 | hip.FLAG_F32_ACTIVATION), (1, hip.FLAG_VECTOR_128), (2, hip.FLAG_VECTOR_128),
                      (8, hip.FLAG_VECTOR_128 | hip.FLAG_F32_ACTIVATION)]:
        m = pkg.synth.make_numpy(pkg.synth.CONFIGS["tiny-llama-tied" if wt == 2 else "mid-llama"], wtype=wt, seed=3)
        with pytest.raises(hip.Gl3Error) as e:
            plan_mod.HipMasterPlan(m, flags=flags)
        assert e.value.code == -2, (wt, flags)
    with pytest.raises(hip.Gl3Error) as e:
        plan_mod.HipMasterPlan(pkg.synth.make_numpy(pkg.synth.CONFIGS["tiny-llama"], wtype=1, seed=3), flags=hip.FLAG_VECTOR_512 | hip.FLAG_SCALAR_DOT)
    assert e.value.code == -1
    m = pkg.synth.make_numpy(pkg.synth.CONFIGS["tiny-llama"], wtype=8, seed=3)
    plan = plan_mod.HipMasterPlan(m, flags=hip.FLAG_VECTOR_512)                 # dotQ8Activation is scalar: nothing changes
    ref = plan_mod.HipMasterPlan(m)
    assert np.array_equal(plan.forward_decode(5, 0), ref.forward_decode(5, 0))
    plan.freeTornadoExecutionPlan(); ref.freeTornadoExecutionPlan()


@pytest.mark.parametrize("cfg", ["mid-llama", "mid-qwen3", "tiny-llama-tied", "mid-qwen2", "mid-granite", "mid-phi3"])
def test_q8_0_with_f32_activation_matches_c_oracle_live(pkg, orc, planmod, cfg):
    """SURVEY 8 a4': Q8_0 matrices with -Dllama.quantizeActivation=false (GL3_FLAG_F32_ACTIVATION) = Q8_0FloatTensor.vectorDot on
    the f32 activation, 256-bit species (matvec_vl_kernel<WT_Q8_0>).  Logits, per-layer x, device argmax and the KV cache after a
    token-by-token prefill are bit-identical to the oracle in the same mode."""
    plan_mod, hip = planmod
    m = pkg.synth.make_numpy(pkg.synth.CONFIGS[cfg], wtype=8, seed=29)
    plan = plan_mod.HipMasterPlan(m, prefill_batch_size=16, flags=hip.FLAG_LAYER_TAPS | hip.FLAG_F32_ACTIVATION)
    o = orc.COracle(m, vector_bits=256, f32_activation=True)
    toks = pkg.javarand.bench_tokens(m.cfg.vocab, 12)
    plan.prefill(toks[:5], 0)
    o.prefill(toks[:5], 0)
    for pos in range(5, 12):
        ref, lx = o.forward(toks[pos], pos, layer_x=True)
        got = plan.tornadoVMForwardDecode(toks[pos], pos)
        assert np.array_equal(got, ref), (pos, rel(got, ref))
        for l in range(m.cfg.n_layers):
            assert np.array_equal(plan.layer_x(l), lx[l])
        assert plan.forward_decode_argmax(toks[pos], pos) == orc.argmax(ref)
    for l in range(m.cfg.n_layers):
        k, v = plan.kv(l, 3)
        ko, vo = o.kv(l, 3)
        assert np.array_equal(k, ko) and np.array_equal(v, vo)
    plan.freeTornadoExecutionPlan()


def test_graph_and_eager_launches_agree_bitwise(pkg, orc, planmod):
    plan_mod, hip = planmod
    m = pkg.synth.make_numpy(pkg.synth.CONFIGS["mid-llama"], seed=3)
    a = plan_mod.HipMasterPlan(m)
    b = plan_mod.HipMasterPlan(m, flags=hip.FLAG_NO_GRAPH)
    o = orc.COracle(m)
    for pos, t in enumerate(pkg.javarand.bench_tokens(m.cfg.vocab, 6)):
        ref = o.forward(t, pos)
        assert np.array_equal(a.forward_decode(t, pos), ref) and np.array_equal(b.forward_decode(t, pos), ref)
    # replaying the same position is idempotent (KV row is overwritten with the same values)
    l1 = a.forward_decode(5, 6)
    assert np.array_equal(l1, a.forward_decode(5, 6))
    a.freeTornadoExecutionPlan(); b.freeTornadoExecutionPlan()


@pytest.mark.parametrize("cfg,depths", [("mid-llama", [129, 255, 511, 640, 764]), ("mid-qwen3", [200, 513, 765]), ("mid-qwen2", [300]), ("mid-granite", [257, 700]),
                                        ("mid-phi3", [384, 766]), ("mha-llama", [450])])      # head sizes 64 / 128, kvMul 4 / 6 (pair fallback) / 3 / 1, qk-norm, bias, attention scale
def test_one_launch_attention_between_128_and_767_positions(pkg, orc, planmod, cfg, depths):
    """Positions 128 .. 767: the two-launch pair (scores; softmax + weighted V sum) by default, and the r6 experiment attn_mid_kernel
    (GL3_ATTN_FUSED_MID=1: RoPE, KV write, scores with K streamed through LDS, softmax and the weighted V sum of a kv head's query heads in ONE launch
    per layer — measured slower, kept selectable; kvMul > 4 and other head sizes always take the pair).  Decode steps behind a batched prefill of d
    positions, at depths on both sides of every 128-step K tile edge and at the last position of the regime; logits and the KV rows the steps wrote
    must equal the oracle bit for bit on both forms."""
    plan_mod, hip = planmod
    base = pkg.synth.CONFIGS[cfg]
    m = pkg.synth.make_numpy(pkg.synth.ModelConfig(**{**base.__dict__, "ctx": 776}), seed=31)
    pair = plan_mod.HipMasterPlan.initializeTornadoVMPlan(m, prefill_batch_size=256)
    os.environ["GL3_ATTN_FUSED_MID"] = "1"
    try:
        plan = plan_mod.HipMasterPlan.initializeTornadoVMPlan(m, prefill_batch_size=256)
    finally:
        os.environ.pop("GL3_ATTN_FUSED_MID", None)
    o = orc.COracle(m)
    toks = pkg.javarand.bench_tokens(m.cfg.vocab, 776)
    done = 0
    for d in depths:
        while done < d:                                   # prefill up to depth d in chunks
            c = min(256, d - done)
            for pl in (plan, pair):
                pl.tornadoVMForwardBatchPrefill(toks[done:done + c], done)
            o.prefill(toks[done:done + c], done)
            done += c
        for pos in range(d, min(d + 2, 768)):
            ref = o.forward(toks[pos], pos)
            got = plan.forward_decode(toks[pos], pos)
            assert np.array_equal(got, ref), (cfg, pos, rel(got, ref))
            assert np.array_equal(pair.forward_decode(toks[pos], pos), ref), (cfg, "pair", pos)
            done = pos + 1
        for l in range(m.cfg.n_layers):
            k, v = plan.kv(l, done - 1)
            ko, vo = o.kv(l, done - 1)
            assert np.array_equal(k, ko) and np.array_equal(v, vo), (cfg, l, done - 1)
    plan.freeTornadoExecutionPlan(); pair.freeTornadoExecutionPlan()


@pytest.mark.parametrize("cfg", ["mid-llama", "mid-qwen3", "mid-qwen2", "mha-llama", "mid-granite", "phi3-hs96"])   # head sizes 64 / 128 / 96, kvMul 4 / 6 / 1
def test_fused_short_context_attention_and_the_handover_at_128(pkg, orc, planmod, cfg):
    """Positions < 128 run attn_head_kernel (one launch per layer, one workgroup per query head), later ones the scores +
    softmax/PV pair (positions < 768) or — `deep`, with GL3_ATTN_MID=0 right behind 128 — the four-launch long-context path (scores,
    exp, exact parallel sum, chain-wavefront PV: r5); all must reproduce the oracle bit for bit, also across the handover."""
    plan_mod, hip = planmod
    base = pkg.synth.CONFIGS[cfg]
    m = pkg.synth.make_numpy(pkg.synth.ModelConfig(**{**base.__dict__, "ctx": 160}), seed=29)
    plan = plan_mod.HipMasterPlan(m)
    os.environ["GL3_NO_FUSED_ATTN"] = "1"
    try:
        plain = plan_mod.HipMasterPlan(m)
    finally:
        os.environ.pop("GL3_NO_FUSED_ATTN", None)
    os.environ["GL3_ATTN_MID"] = "0"
    try:
        deep = plan_mod.HipMasterPlan(m)
    finally:
        os.environ.pop("GL3_ATTN_MID", None)
    o = orc.COracle(m)
    toks = pkg.javarand.bench_tokens(m.cfg.vocab, 134)
    for pos in range(122):
        plan.forward_decode(toks[pos], pos, copy=False)
        plain.forward_decode(toks[pos], pos, copy=False)
        deep.forward_decode(toks[pos], pos, copy=False)
    o.prefill(toks[:122], 0)
    for pos in range(122, 134):
        ref = o.forward(toks[pos], pos)
        got = plan.forward_decode(toks[pos], pos)
        assert np.array_equal(got, ref), (pos, rel(got, ref))
        assert np.array_equal(plain.forward_decode(toks[pos], pos), ref), pos
        got = deep.forward_decode(toks[pos], pos)
        assert np.array_equal(got, ref), ("long-context path", pos, rel(got, ref))
    for l in range(m.cfg.n_layers):
        for p in (0, 60, 127, 128, 133):
            k, v = plan.kv(l, p)
            ko, vo = o.kv(l, p)
            assert np.array_equal(k, ko) and np.array_equal(v, vo)
    plan.freeTornadoExecutionPlan(); plain.freeTornadoExecutionPlan(); deep.freeTornadoExecutionPlan()


def test_sequential_prefill_then_decode(pkg, orc, planmod):
    plan_mod, hip = planmod
    m = pkg.synth.make_numpy(pkg.synth.CONFIGS["tiny-llama"], seed=7)
    plan = plan_mod.HipMasterPlan(m)           # prefill_batch_size = 1 -> tornadoVMForwardPrefill semantics
    o = orc.COracle(m)
    toks = pkg.javarand.bench_tokens(m.cfg.vocab, 12)
    plan.prefill(toks[:11], 0)
    o.prefill(toks[:11], 0)
    assert np.array_equal(plan.tornadoVMForwardDecode(toks[11], 11), o.forward(toks[11], 11))
    plan.freeTornadoExecutionPlan()


def test_error_behaviour(pkg, planmod):
    plan_mod, hip = planmod
    m = pkg.synth.make_numpy(pkg.synth.CONFIGS["tiny-llama"], seed=7)
    plan = plan_mod.HipMasterPlan(m)
    with pytest.raises(hip.Gl3Error) as e:
        plan.forward_decode(m.cfg.vocab, 0)                   # token out of range
    assert e.value.code == -1
    with pytest.raises(hip.Gl3Error):
        plan.forward_decode(1, m.cfg.ctx)                     # beyond the KV cache
    with pytest.raises(hip.Gl3Error):
        plan.layer_x(0)                                       # taps not enabled
    plan.freeTornadoExecutionPlan()
    m4 = pkg.synth.make_numpy(pkg.synth.CONFIGS["tiny-llama"], wtype=0, seed=7)
    with pytest.raises(hip.Gl3Error) as e:
        plan_mod.HipMasterPlan(m4)                            # F32 matrices: GL3_E_UNSUPPORTED (Q8_0 / F16 / Q4_0 only)
    assert e.value.code == -2
    m5 = pkg.synth.make_numpy(pkg.synth.CONFIGS["tiny-llama"], wtype=2, seed=7)
    plan = plan_mod.HipMasterPlan(m5, prefill_batch_size=8, n_seqs=2, flags=hip.FLAG_SCALAR_DOT)
    with pytest.raises(hip.Gl3Error) as e:                    # the scalar dot order has no batched path (one K-long chain per row)
        plan.forward_decode_batch([1, 2], [0, 1], [0, 0])
    assert e.value.code == -2
    plan.freeTornadoExecutionPlan()
    # an inner dimension the vector-order matvecs cannot stage in LDS is refused at gl3_create (before any allocation), not at the first launch
    import ctypes as C
    d = hip.ModelDesc(struct_size=C.sizeof(hip.ModelDesc), arch=0, dim=4096, hidden=16640, n_layers=1, n_heads=32, n_kv_heads=8, head_size=128, vocab=1024, ctx=64,
                      rms_eps=1e-5, weight_type=1, max_batch=1, device=0, tp_rank=0, tp_size=1, flags=0, n_seqs=1, embedding_scale=1.0, attention_scale=0.0,
                      residual_scale=1.0, logit_scale=1.0)
    h = C.c_void_p()
    assert hip.lib().gl3_create(C.byref(d), C.byref(h)) == -2 and not h.value
    assert b"16384" in hip.lib().gl3_last_error(None)
    d.weight_type = 8                                         # Q8_0 with the int8 activation has no such limit
    assert hip.lib().gl3_create(C.byref(d), C.byref(h)) == 0
    hip.lib().gl3_destroy(h)


@pytest.mark.parametrize("cfg,batch,chunks", [("tiny-llama", 8, [8, 8, 5]), ("mid-llama", 64, [40, 64, 3]), ("mid-qwen3", 32, [30, 7]), ("mid-qwen2", 64, [50, 9]), ("mid-granite", 64, [33, 20]), ("mid-phi3", 64, [41, 6]), ("mid-devstral", 64, [45, 18]),
                                             ("tiny-llama-tied", 512, [37]), ("phi3-hs96", 64, [50, 14])])
def test_batched_prefill_is_bit_identical_to_the_cpu_path(pkg, orc, planmod, cfg, batch, chunks):
    """tornadoVMForwardBatchPrefill (MFMA int8 GEMM path) vs batchForwardJavaPrefill: same KV cache, same x of the
    last token, and the decode step that follows returns the same logits — all bit for bit.  Ragged chunk sizes
    (not multiples of 32 / 64 / 256) and chunks that start at a non-zero position are covered."""
    plan_mod, hip = planmod
    m = pkg.synth.make_numpy(pkg.synth.CONFIGS[cfg], seed=33)
    plan = plan_mod.HipMasterPlan.initializeTornadoVMPlan(m, prefill_batch_size=batch)
    o = orc.COracle(m)
    n = sum(chunks)
    toks = pkg.javarand.bench_tokens(m.cfg.vocab, n + 2)
    pos = 0
    for c in chunks:
        plan.tornadoVMForwardBatchPrefill(toks[pos:pos + c], pos)
        o.prefill(toks[pos:pos + c], pos)
        pos += c
        assert np.array_equal(plan.x(), o.x())
    for l in range(m.cfg.n_layers):
        for p in (0, 1, n // 2, n - 1):
            k, v = plan.kv(l, p)
            ko, vo = o.kv(l, p)
            assert np.array_equal(k, ko) and np.array_equal(v, vo), (l, p)
    for i in range(2):
        assert np.array_equal(plan.tornadoVMForwardDecode(toks[n + i], n + i), o.forward(toks[n + i], n + i))
    with pytest.raises(hip.Gl3Error):
        plan.tornadoVMForwardBatchPrefill(toks[:batch + 1] if batch + 1 <= len(toks) else list(toks) * 300, 0)   # chunk > max_batch
    plan.freeTornadoExecutionPlan()


@pytest.mark.parametrize("cfg,batch,chunks", [("mid-llama", 128, [100, 57]), ("mid-qwen2", 160, [129, 20]), ("mid-phi3", 128, [65, 70]), ("phi3-hs96", 160, [150]),
                                             ("mid-granite", 96, [96, 50]), ("mid-qwen3", 36, [36]), ("ragged-llama", 192, [131, 66]), ("ragged-llama", 70, [70, 65, 64]),
                                             # kvMul 4 with head sizes 128 / 64: the one-launch attention with its products on the matrix pipe (pf_attn_fused3_kernel),
                                             # chunks that end inside an 8-token tile and start at a non-zero position
                                             ("mid-devstral", 128, [101, 58]), ("tiny-devstral", 64, [37, 26])])
def test_batched_prefill_chunks_above_64_tokens(pkg, orc, planmod, cfg, batch, chunks):
    """Chunks of more than 64 tokens take the LDS-tiled GEMM (r6: pf_gemm3_kernel — 128 x 128, 96 x 128 and 64 x 128 workgroup tiles picked by
    the matrix's row count): ragged token counts (the last 128-token tile partly empty), row counts that are no multiple of a tile, K = 9 / 27
    blocks (the last K stage holds one real block: zero scale operands for the padded ones), a chunk that starts at a non-zero position."""
    plan_mod, hip = planmod
    m = pkg.synth.make_numpy(pkg.synth.CONFIGS[cfg], seed=35)
    plan = plan_mod.HipMasterPlan.initializeTornadoVMPlan(m, prefill_batch_size=batch)
    o = orc.COracle(m)
    n = sum(chunks)
    toks = pkg.javarand.bench_tokens(m.cfg.vocab, n + 1)
    pos = 0
    for c in chunks:
        plan.tornadoVMForwardBatchPrefill(toks[pos:pos + c], pos)
        o.prefill(toks[pos:pos + c], pos)
        pos += c
        assert np.array_equal(plan.x(), o.x())
    for l in range(m.cfg.n_layers):
        for p in (0, min(63, n - 1), min(64, n - 1), n // 2, n - 1):
            k, v = plan.kv(l, p)
            ko, vo = o.kv(l, p)
            assert np.array_equal(k, ko) and np.array_equal(v, vo), (l, p)
    assert np.array_equal(plan.tornadoVMForwardDecode(toks[n], n), o.forward(toks[n], n))
    plan.freeTornadoExecutionPlan()


@pytest.mark.parametrize("cfg,wtype,f32act,batch,chunks", [("mid-llama", 1, False, 64, [40, 23, 3]), ("mid-llama", 2, False, 64, [40, 23, 3]),
                                                          ("mid-qwen3", 1, False, 32, [30, 7]), ("mid-qwen3", 2, False, 32, [17, 20]),
                                                          ("mid-llama", 8, True, 64, [33, 31]), ("mid-qwen2", 8, True, 16, [16, 5]),
                                                          ("tiny-llama-tied", 1, False, 512, [37]), ("mid-granite", 2, False, 64, [35, 2]),
                                                          ("mid-phi3", 1, False, 64, [64, 1])])
def test_batched_prefill_of_the_f32_activation_types(pkg, orc, planmod, cfg, wtype, f32act, batch, chunks):
    """SURVEY 8 a15 for F16 / Q4_0 / Q8_0-with-f32-activation: tornadoVMForwardBatchPrefill on the Vector-API-order GEMMs
    (gl3_prefill_vl.h: f32 MFMA FMA chains for F16, VALU for the block formats) vs batchForwardJavaPrefill of the oracle in the same
    mode: x of the last token, KV rows and the decode steps that follow, bit for bit; ragged chunks at non-zero start positions."""
    plan_mod, hip = planmod
    m = pkg.synth.make_numpy(pkg.synth.CONFIGS[cfg], wtype=wtype, seed=35)
    plan = plan_mod.HipMasterPlan.initializeTornadoVMPlan(m, prefill_batch_size=batch, flags=hip.FLAG_F32_ACTIVATION if f32act else 0)
    o = orc.COracle(m, vector_bits=256, f32_activation=f32act)
    n = sum(chunks)
    toks = pkg.javarand.bench_tokens(m.cfg.vocab, n + 2)
    pos = 0
    for c in chunks:
        plan.tornadoVMForwardBatchPrefill(toks[pos:pos + c], pos)
        o.prefill(toks[pos:pos + c], pos)
        pos += c
        assert np.array_equal(plan.x(), o.x()), (pos, rel(plan.x(), o.x()))
    for l in range(m.cfg.n_layers):
        for p in (0, 1, n // 2, n - 1):
            k, v = plan.kv(l, p)
            ko, vo = o.kv(l, p)
            assert np.array_equal(k, ko) and np.array_equal(v, vo), (l, p)
    for i in range(2):
        assert np.array_equal(plan.tornadoVMForwardDecode(toks[n + i], n + i), o.forward(toks[n + i], n + i))
    plan.freeTornadoExecutionPlan()


@pytest.mark.parametrize("cfg,wtype,f32act,nseq", [("mid-llama", 1, False, 3), ("mid-llama", 2, False, 5), ("mid-qwen3", 8, True, 3), ("tiny-llama-tied", 1, False, 17)])
def test_static_batched_decode_of_the_f32_activation_types(pkg, orc, planmod, cfg, wtype, f32act, nseq):
    """gl3_forward_decode_batch for F16 / Q4_0 / Q8_0-with-f32-activation: n sequences advance one token per step through the
    Vector-API-order GEMMs; logits and greedy ids of every sequence equal its own oracle run."""
    plan_mod, hip = planmod
    m = pkg.synth.make_numpy(pkg.synth.CONFIGS[cfg], wtype=wtype, seed=45)
    plan = plan_mod.HipMasterPlan(m, prefill_batch_size=max(16, nseq), n_seqs=nseq, flags=hip.FLAG_F32_ACTIVATION if f32act else 0)
    oracles = [orc.COracle(m, vector_bits=256, f32_activation=f32act) for _ in range(nseq)]
    rng = np.random.default_rng(5)
    lens = [2 + (i % 4) for i in range(nseq)]
    for s_ in range(nseq):
        prompt = rng.integers(0, m.cfg.vocab, lens[s_]).tolist()
        plan.prefill_seq(s_, prompt, 0)
        oracles[s_].prefill(prompt, 0)
    cur = [int(rng.integers(0, m.cfg.vocab)) for _ in range(nseq)]
    pos = list(lens)
    for step in range(3):
        order = list(range(nseq))
        if step % 2:
            order.reverse()
        logits, ids = plan.forward_decode_batch([cur[s_] for s_ in order], order, [pos[s_] for s_ in order])
        for row, s_ in enumerate(order):
            ref = oracles[s_].forward(cur[s_], pos[s_])
            assert np.array_equal(logits[row], ref), (step, s_)
            assert ids[row] == orc.argmax(ref)
            cur[s_], pos[s_] = int(ids[row]), pos[s_] + 1
    plan.freeTornadoExecutionPlan()


@pytest.mark.parametrize("cfg,nseq", [("tiny-qwen3", 5), ("mid-llama", 3), ("mid-granite", 4), ("phi3-hs96", 3)])
def test_static_batched_decode_matches_independent_cpu_runs(pkg, orc, planmod, cfg, nseq):
    """BASELINE config 5 shape of work: n independent sequences (own KV caches, different prompt lengths) advance one
    token per step through ONE batched GEMM pass; every sequence's logits must equal its own CPU run bit for bit."""
    plan_mod, hip = planmod
    m = pkg.synth.make_numpy(pkg.synth.CONFIGS[cfg], seed=44)
    plan = plan_mod.HipMasterPlan(m, prefill_batch_size=16, n_seqs=nseq)
    oracles = [orc.COracle(m) for _ in range(nseq)]
    rng = np.random.default_rng(3)
    lens = [3 + 2 * i for i in range(nseq)]
    prompts = [rng.integers(0, m.cfg.vocab, n).tolist() for n in lens]
    for s in range(nseq):
        plan.prefill_seq(s, prompts[s], 0)
        oracles[s].prefill(prompts[s], 0)
    cur = [int(rng.integers(0, m.cfg.vocab)) for _ in range(nseq)]
    pos = list(lens)
    for step in range(4):
        order = list(range(nseq))
        if step % 2:
            order.reverse()                      # batch row order is arbitrary
        logits, ids = plan.forward_decode_batch([cur[s] for s in order], order, [pos[s] for s in order])
        for row, s in enumerate(order):
            ref = oracles[s].forward(cur[s], pos[s])
            assert np.array_equal(logits[row], ref), (step, s)
            assert ids[row] == orc.argmax(ref)
            cur[s], pos[s] = int(ids[row]), pos[s] + 1          # greedy continuation per sequence
    k, v = plan.kv_seq(nseq - 1, 0, lens[-1])
    ko, vo = oracles[-1].kv(0, lens[-1])
    assert np.array_equal(k, ko) and np.array_equal(v, vo)
    with pytest.raises(hip.Gl3Error):
        plan.forward_decode_batch([1, 2], [0, 0], [pos[0], pos[0]])       # duplicate sequence id
    plan.freeTornadoExecutionPlan()


@pytest.mark.parametrize("cfg,wtype", [("tiny-llama", 8), ("tiny-qwen3", 8), ("tiny-llama-tied", 1), ("tiny-qwen2", 8), ("tiny-granite", 8), ("tiny-phi3", 8),
                                       ("tiny-devstral", 8)])          # "mistral3" file: key_length head size + YaRN table built by the library
def test_native_gguf_loader_builds_the_same_plan(pkg, orc, planmod, tmp_path, cfg, wtype):
    """gl3_load_gguf (mmap + native config / tensor-name map / RoPE table) vs the per-tensor upload path driven from Python."""
    plan_mod, hip = planmod
    m = pkg.synth.make_numpy(pkg.synth.CONFIGS[cfg], wtype=wtype, seed=13)
    path = str(tmp_path / "m.gguf")
    m.write_gguf(path)
    a = plan_mod.HipMasterPlan(m)
    b = plan_mod.HipMasterPlan.from_gguf(path, prefill_batch_size=8)
    assert (b.cfg.dim, b.cfg.n_layers, b.cfg.vocab, b.cfg.head_size) == (m.cfg.dim, m.cfg.n_layers, m.cfg.vocab, m.cfg.head_size)
    o = orc.COracle(pkg.synth.SynthModel.from_gguf(path), vector_bits=0 if wtype == 8 else 256)      # the oracle reads the SAME file through the Python GGUF reader
    toks = pkg.javarand.bench_tokens(m.cfg.vocab, 10)
    b.prefill(toks[:5], 0)
    for pos, t in enumerate(toks):
        ref = o.forward(t, pos)
        assert np.array_equal(a.forward_decode(t, pos), ref), pos
        if pos >= 5:
            assert np.array_equal(b.forward_decode(t, pos), ref), pos
    a.freeTornadoExecutionPlan(); b.freeTornadoExecutionPlan()
    with pytest.raises(hip.Gl3Error):
        plan_mod.HipMasterPlan.from_gguf(str(tmp_path / "missing.gguf"))


def test_native_bench_host_over_the_c_abi(pkg, orc, planmod, tmp_path):
    """tools/gl3_bench (plain C++, links only the C-ABI): loads a GGUF natively, runs the LlamaBench protocol (test names, -b, -d, -pg,
    output formats of J/bench/LlamaBench.java:52-59,99-109,309-372) and must produce the same greedy ids as the CPU oracle for the
    java.util.Random(42) token stream — at depth 0 and behind an untimed prefill of -d positions (ids indexed by absolute position)."""
    import json
    import subprocess
    plan_mod, hip = planmod
    m = pkg.synth.make_numpy(pkg.synth.CONFIGS["tiny-llama"], seed=23)
    path = str(tmp_path / "tiny.gguf")
    m.write_gguf(path)
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "gl3_bench")
    out = subprocess.run([exe, "-m", path, "-p", "16", "-n", "12", "-b", "8", "-r", "1", "--ids"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "| tiny | Q8_0 |" in out.stdout and "| pp16 b8 |" in out.stdout and "| tg12 b8 |" in out.stdout
    ids = [int(x) for x in out.stderr.split("greedy ids:")[1].split("\n")[0].split()]
    o = orc.COracle(pkg.synth.SynthModel.from_gguf(path))
    toks = pkg.javarand.bench_tokens(m.cfg.vocab, 40)
    assert ids == [orc.argmax(o.forward(toks[i], i)) for i in range(12)]          # greedy ids of the CPU oracle on the same file
    # -d 20: 20 positions prefilled untimed (chunks of 8), then tg at positions 20 ..; -pg; json output with one row per test
    out = subprocess.run([exe, "-m", path, "-n", "6", "-pg", "8,4", "-d", "0,20", "-b", "8", "-r", "2", "-o", "json", "--ids"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    rows = json.loads(out.stdout)
    assert [r["test"] for r in rows] == ["tg6 b8", "pp8+tg4 b8", "tg6@d20 b8", "pp8+tg4@d20 b8"]
    assert all(r["model"] == "tiny" and r["quant"] == "Q8_0" and len(r["samples_ts"]) == 2 and r["avg_ts"] > 0 for r in rows)
    id_lines = [[int(x) for x in l.split(":")[1].split()] for l in out.stderr.splitlines() if l.startswith("greedy ids:")]
    o2 = orc.COracle(pkg.synth.SynthModel.from_gguf(path))
    o2.prefill(toks[:20], 0)
    assert id_lines[2] == [orc.argmax(o2.forward(toks[20 + i], 20 + i)) for i in range(6)]      # tg6@d20
    for fmt, needle in (("csv", "model,quant,size_gib,params_b,backend,test,avg_ts,stddev_ts,samples"), ("sql", "INSERT INTO llama_bench VALUES ('tiny', 'Q8_0',"),
                        ("jsonl", '{"model": "tiny", "quant": "Q8_0",')):
        out = subprocess.run([exe, "-m", path, "-p", "8", "-n", "0", "-r", "1", "--no-warmup", "-o", fmt], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0 and needle in out.stdout, (fmt, out.stdout, out.stderr)


@pytest.mark.parametrize("cfg,window", [("tiny-llama", None), ("tiny-qwen3", None), ("tiny-llama", 1024), ("phi3-hs96", 1024)])
def test_long_context_prefill_and_decode(pkg, orc, planmod, cfg, window, monkeypatch):
    """Context beyond one V slab (PV_ROWS = 1024) and beyond 16 score tiles: batched prefill in 512-token chunks that start
    at non-zero positions, then decode steps at positions > 1024, all bit-identical to the oracle.
    window = 1024 (GL3_ATTN_WINDOW): the softmax rows of positions >= 1024 no longer fit the LDS window and run in windows with the
    sequential sum carried across them — the path contexts beyond 16 k positions take (decode kernel; phi3-hs96 = head size 96 also
    takes the per-token prefill kernels)."""
    plan_mod, hip = planmod
    if window:
        monkeypatch.setenv("GL3_ATTN_WINDOW", str(window))
    base = pkg.synth.CONFIGS[cfg]
    m = pkg.synth.make_numpy(pkg.synth.ModelConfig(**{**base.__dict__, "ctx": 1300}), seed=31)
    plan = plan_mod.HipMasterPlan.initializeTornadoVMPlan(m, prefill_batch_size=512)
    o = orc.COracle(m)
    toks = pkg.javarand.bench_tokens(m.cfg.vocab, 1104)
    pos = 0
    for c in (512, 512, 76):
        plan.tornadoVMForwardBatchPrefill(toks[pos:pos + c], pos)
        pos += c
    o.prefill(toks[:1100], 0)
    for p in range(1100, 1104):
        ref = o.forward(toks[p], p)
        got = plan.forward_decode(toks[p], p)
        assert np.array_equal(got, ref), (p, rel(got, ref))
    for l in range(m.cfg.n_layers):
        for p in (0, 511, 512, 1023, 1024, 1099, 1103):
            k, v = plan.kv(l, p)
            ko, vo = o.kv(l, p)
            assert np.array_equal(k, ko) and np.array_equal(v, vo), (l, p)
    plan.freeTornadoExecutionPlan()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["tiny-llama", "tiny-qwen3", "mid-qwen3"])      # head sizes 32 (one-tile scores kernel), 64 and 128 (looping scores kernel, ~10 tiles per workgroup)
def test_context_beyond_20k_positions(pkg, orc, planmod, cfg):
    """Round 2 rejected contexts above ~20 k positions (the decode attention kept a ctx-long softmax row in LDS); the reference has
    no such cap (InferenceCore.java:98-137 is O(pos)).  A decode step at position 20 600 of a 21 000-position context, on a KV cache
    that holds one prefilled chunk and zeros elsewhere (both sides zero-initialise it), is bit-identical to the oracle."""
    plan_mod, hip = planmod
    base = pkg.synth.CONFIGS[cfg]
    m = pkg.synth.make_numpy(pkg.synth.ModelConfig(**{**base.__dict__, "ctx": 21000}), seed=5)
    plan = plan_mod.HipMasterPlan.initializeTornadoVMPlan(m, prefill_batch_size=64)
    o = orc.COracle(m)
    toks = pkg.javarand.bench_tokens(m.cfg.vocab, 40)
    plan.tornadoVMForwardBatchPrefill(toks[:32], 0)
    o.prefill(toks[:32], 0)
    for i, p in enumerate((20600, 20601, 16384, 16383)):
        ref = o.forward(toks[32 + i], p)
        got = plan.forward_decode(toks[32 + i], p)
        assert np.array_equal(got, ref), (p, rel(got, ref))
