"""Qwen2-MoE / Qwen1.5-MoE decode (GL3_ARCH_QWEN2MOE) against the CPU oracle of InferenceCore.forwardJavaQwen2MoE (:263-422).

Everything is np.array_equal: logits, x after every layer, the KV cache, the device argmax — and the routing decision of the last
layer itself (expert ids, their probabilities, the shared expert's sigmoid gate), read back through gl3_get_buffer 7 / 8.
"""
import ctypes as C
import os

import numpy as np
import pytest

import __graft_entry__ as ge

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def planmod():
    from importlib import import_module
    ge.load_package()
    return import_module(ge.PKG_NAME + ".plan"), import_module(ge.PKG_NAME + ".hip")


def _routing(plan):
    k = plan.cfg.n_experts_used
    w = plan.buffer(7, k + 1)
    return plan.buffer(8, k).astype(np.int32), w[:k], w[k]


def test_moe_decode_matches_golden_fixture(pkg, planmod):
    plan_mod, hip = planmod
    g = np.load(os.path.join(GOLD, "tiny_qwen2moe_q8_0.npz"))
    m = pkg.synth.make_numpy(pkg.synth.CONFIGS["tiny-qwen2moe"], wtype=8, seed=31)
    plan = plan_mod.HipMasterPlan(m, flags=hip.FLAG_LAYER_TAPS)
    toks, n_prompt = g["tokens"], len(g["prompt"])
    for pos in range(g["logits"].shape[0]):
        lg = plan.tornadoVMForwardDecode(int(toks[pos]), pos)
        assert np.array_equal(lg, g["logits"][pos]), pos
        if pos >= n_prompt - 1:
            assert int(np.argmax(lg)) == toks[pos + 1], pos
    for l in range(m.cfg.n_layers):
        assert np.array_equal(plan.layer_x(l), g["last_layer_x"][l])
        k, v = plan.kv(l, g["logits"].shape[0] - 1)
        assert np.array_equal(k, g["k_last"][l]) and np.array_equal(v, g["v_last"][l])
    plan.freeTornadoExecutionPlan()


@pytest.mark.parametrize("cfg,graph", [("tiny-qwen2moe", True), ("mid-qwen2moe", True), ("mid-qwen2moe", False)])
def test_moe_decode_matches_c_oracle_live(pkg, orc, planmod, cfg, graph):
    """60 experts / top-4 (the Qwen1.5-MoE-A2.7B routing shape) and 8 / top-2; graph replay and eager launches.  The expert ids the
    device picked are compared too: a wrong pick with a near-equal probability could hide in the logits' low bits."""
    plan_mod, hip = planmod
    m = pkg.synth.make_numpy(pkg.synth.CONFIGS[cfg], seed=21)
    plan = plan_mod.HipMasterPlan(m, flags=hip.FLAG_LAYER_TAPS | (0 if graph else hip.FLAG_NO_GRAPH))
    o = orc.COracle(m)
    toks = pkg.javarand.bench_tokens(m.cfg.vocab, 16)
    picked = set()
    for pos, t in enumerate(toks):
        ref, lx = o.forward(t, pos, layer_x=True)
        got = plan.tornadoVMForwardDecode(t, pos)
        assert np.array_equal(got, ref), pos
        for l in range(m.cfg.n_layers):
            assert np.array_equal(plan.layer_x(l), lx[l]), (pos, l)
        sel, w, sw = _routing(plan)
        rsel, rw, rsw = o.moe_routing()
        assert sel.tolist() == rsel.tolist() and np.array_equal(w, rw) and sw == rsw, pos
        picked.update(sel.tolist())
        assert plan.forward_decode_argmax(t, pos) == orc.argmax(ref)
    assert len(picked) > m.cfg.n_experts_used           # the routing moved between tokens
    plan.freeTornadoExecutionPlan()


def test_moe_sequential_prefill_then_decode(pkg, orc, planmod):
    """tornadoVMForwardPrefill token by token (the only prefill of this family, as in the reference) leaves the oracle's KV cache."""
    plan_mod, hip = planmod
    m = pkg.synth.make_numpy(pkg.synth.CONFIGS["mid-qwen2moe"], seed=5)
    plan = plan_mod.HipMasterPlan.initializeTornadoVMPlan(m, prefill_batch_size=16)      # accepted; chunks of 16 + 4 run token by token
    o = orc.COracle(m)
    toks = pkg.javarand.bench_tokens(m.cfg.vocab, 24)
    plan.prefill(toks[:20], 0)
    o.prefill(toks[:20], 0)
    assert np.array_equal(plan.x(), o.x())
    for l in range(m.cfg.n_layers):
        for p in (0, 7, 19):
            k, v = plan.kv(l, p)
            ko, vo = o.kv(l, p)
            assert np.array_equal(k, ko) and np.array_equal(v, vo), (l, p)
    for pos in range(20, 24):
        assert np.array_equal(plan.tornadoVMForwardDecode(toks[pos], pos), o.forward(toks[pos], pos)), pos
    plan.freeTornadoExecutionPlan()


def test_moe_layer_at_the_a2_7b_shapes(pkg, orc, planmod):
    """One layer at Qwen1.5-MoE-A2.7B's own sizes: dim 2048, 16 / 16 heads of 128, 60 experts of 1408 (88 strips, K = 1408 = 11 tile
    groups for the down projections), top-4, shared expert 5632."""
    import torch
    plan_mod, hip = planmod
    m = pkg.synth.make_torch(pkg.synth.CONFIGS["a2.7b-moe-layer"], wtype=8, seed=9, device="cuda" if torch.cuda.is_available() else "cpu")
    plan = plan_mod.HipMasterPlan(m, flags=hip.FLAG_LAYER_TAPS)
    o = orc.COracle(m)
    toks = pkg.javarand.bench_tokens(m.cfg.vocab, 6)
    for pos, t in enumerate(toks):
        ref, lx = o.forward(t, pos, layer_x=True)
        assert np.array_equal(plan.tornadoVMForwardDecode(t, pos), ref), pos
        assert np.array_equal(plan.layer_x(0), lx[0]), pos
        sel, w, sw = _routing(plan)
        rsel, rw, rsw = o.moe_routing()
        assert sel.tolist() == rsel.tolist() and np.array_equal(w, rw) and sw == rsw
    plan.freeTornadoExecutionPlan()


def test_moe_native_gguf_loader(pkg, orc, planmod, tmp_path):
    """gl3_load_gguf on a "qwen2moe" file: 3-D expert stacks, F32 router, shared expert in the dense FFN slots."""
    plan_mod, hip = planmod
    m = pkg.synth.make_numpy(pkg.synth.CONFIGS["tiny-qwen2moe"], seed=13)
    path = str(tmp_path / "moe.gguf")
    m.write_gguf(path)
    b = plan_mod.HipMasterPlan.from_gguf(path)
    assert (b.cfg.arch, b.cfg.n_experts, b.cfg.n_experts_used, b.cfg.moe_hidden) == (5, 8, 2, 128)
    o = orc.COracle(pkg.synth.SynthModel.from_gguf(path))
    for pos, t in enumerate(pkg.javarand.bench_tokens(m.cfg.vocab, 8)):
        assert np.array_equal(b.forward_decode(t, pos), o.forward(t, pos)), pos
    b.freeTornadoExecutionPlan()


def test_moe_plan_limits_are_reported_at_create(pkg, planmod):
    """What is not built for this family is refused by gl3_create with GL3_E_UNSUPPORTED, not discovered at the first step."""
    plan_mod, hip = planmod
    c = pkg.synth.CONFIGS["tiny-qwen2moe"]

    def desc(**over):
        d = hip.ModelDesc(C.sizeof(hip.ModelDesc), c.arch, c.dim, c.hidden, c.n_layers, c.n_heads, c.n_kv_heads, c.head_size, c.vocab, c.ctx,
                          c.rms_eps, 8, 1, 0, 0, 1, 0, 1, 1.0, 0.0, 1.0, 1.0, c.n_experts, c.n_experts_used, c.moe_hidden)
        for k, v in over.items():
            setattr(d, k, v)
        return d
    h = C.c_void_p()
    L = hip.lib()
    assert L.gl3_create(C.byref(desc()), C.byref(h)) == 0
    L.gl3_destroy(h)
    for over in (dict(n_seqs=2), dict(tp_size=2), dict(weight_type=1), dict(flags=hip.FLAG_F32_ACTIVATION),
                 dict(dim=8192, n_heads=64, n_kv_heads=16, head_size=128)):       # the router's LDS staging (8 rows of products) stops near dim 4500
        assert L.gl3_create(C.byref(desc(**over)), C.byref(h)) == -2, over
    for over in (dict(n_experts_used=0), dict(n_experts_used=9), dict(moe_hidden=48)):
        assert L.gl3_create(C.byref(desc(**over)), C.byref(h)) == -1, over
    d = desc(arch=2, n_experts=0, n_experts_used=0, moe_hidden=0)          # a dense plan must not carry expert counts
    assert L.gl3_create(C.byref(d), C.byref(h)) == 0
    L.gl3_destroy(h)
    assert L.gl3_create(C.byref(desc(arch=2)), C.byref(h)) == -1
