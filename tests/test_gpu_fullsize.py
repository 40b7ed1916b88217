"""Parity at the SHAPES of the BASELINE.json configurations (the round-1 tests stopped at "mid" shapes):

* one Llama-3-8B-shaped layer (dim 4096, hidden 14336, 32 / 8 heads, head_size 128): K = 14336 gives 112 tile groups per
  strip and exactly 14 * 256 activation quads in the matvec prologue; a 512-token batched prefill (the pp512 -b 512 shape:
  128-token GEMM tiles, 8-wavefront wo / down variant) and decode steps on both attention paths;
* the 128256-row vocabulary projection on dim 4096 (501 strips per workgroup slot, full logits compared);
* BASELINE configs[4]: static-batched decode on two Qwen3-4B-shaped layers (K = 2560 -> 80 blocks, ragged tile group;
  head_size 128 != dim / heads; tied wcls) at B = 32 (bd_gemm_kernel with every column live), B = 33 and B = 64
  (the 32-token-tile pf_gemm variant) against one independent CPU oracle per sequence.

All comparisons are np.array_equal against oracle/gl3_oracle.c on the same weights.
"""
import numpy as np
import pytest

import __graft_entry__ as ge

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def planmod():
    from importlib import import_module
    ge.load_package()
    return import_module(ge.PKG_NAME + ".plan"), import_module(ge.PKG_NAME + ".hip")


def _model(pkg, name, seed, wtype=8, **over):
    """Weights are generated on the GPU (seconds for 230 M parameters) and handed to BOTH the plan and the oracle."""
    import torch
    cfg = pkg.synth.CONFIGS[name]
    if over:
        cfg = pkg.synth.ModelConfig(**{**cfg.__dict__, **over})
    return pkg.synth.make_torch(cfg, wtype=wtype, seed=seed, device="cuda" if torch.cuda.is_available() else "cpu")


def test_llama3_8b_shaped_layer_prefill512_and_decode(pkg, orc, planmod):
    plan_mod, hip = planmod
    m = _model(pkg, "8b-layer", 101)
    toks = pkg.javarand.bench_tokens(m.cfg.vocab, 518)
    # ---- decode from position 0 (attn_head_kernel path), logits + per-layer x + device argmax
    plan = plan_mod.HipMasterPlan.initializeTornadoVMPlan(m, prefill_batch_size=512, flags=hip.FLAG_LAYER_TAPS)
    o = orc.COracle(m)
    for pos in range(5):
        ref, lx = o.forward(toks[pos], pos, layer_x=True)
        got = plan.tornadoVMForwardDecode(toks[pos], pos)
        assert np.array_equal(got, ref), pos
        assert np.array_equal(plan.layer_x(0), lx[0]), pos
        assert plan.forward_decode_argmax(toks[pos], pos) == orc.argmax(ref)
    # ---- pp512 -b 512 shape: one 512-token chunk, then decode at depth 512 (scores + softmax/PV pair, 9 score tiles)
    plan.reset_kv()
    o2 = orc.COracle(m)
    plan.tornadoVMForwardBatchPrefill(toks[:512], 0)
    o2.prefill(toks[:512], 0)
    assert np.array_equal(plan.x(), o2.x())
    for p in (0, 1, 127, 128, 300, 511):
        k, v = plan.kv(0, p)
        ko, vo = o2.kv(0, p)
        assert np.array_equal(k, ko) and np.array_equal(v, vo), p
    for pos in range(512, 516):
        ref = o2.forward(toks[pos], pos)
        assert np.array_equal(plan.tornadoVMForwardDecode(toks[pos], pos), ref), pos
    # ---- a ragged second chunk on top (non-zero start, 128-token tile with 2 live tokens)
    plan.tornadoVMForwardBatchPrefill(toks[516:518], 516)
    o2.prefill(toks[516:518], 516)
    assert np.array_equal(plan.x(), o2.x())
    plan.freeTornadoExecutionPlan()


def test_llama3_8b_shaped_layer_prefill_chunks_behind_1000_positions(pkg, orc, planmod):
    """pp512 @ depth: chunks whose score rows no longer fit the one-launch prefill attention (from ~640 positions at kvMul 4, head size 128)
    take pf_scores_tiled_kernel (tile maxima) -> pf_softmax_rows_kernel (64 / 32 / 16 rows per workgroup for 512- / 256- / 100-token
    chunks, sums as lane-per-row chains) -> pf_pv_tiled_kernel (divides where it stages the weights): residual stream, KV rows and the
    decode steps behind them against the C oracle."""
    plan_mod, hip = planmod
    m = _model(pkg, "8b-layer", 131, ctx=1500)
    toks = pkg.javarand.bench_tokens(m.cfg.vocab, 1400)
    plan = plan_mod.HipMasterPlan.initializeTornadoVMPlan(m, prefill_batch_size=512)
    o = orc.COracle(m)
    pos = 0
    for c in (512, 512, 256, 100):
        plan.tornadoVMForwardBatchPrefill(toks[pos:pos + c], pos)
        o.prefill(toks[pos:pos + c], pos)
        pos += c
        assert np.array_equal(plan.x(), o.x()), pos
    for p in (0, 511, 512, 1023, 1024, 1279, 1280, 1379):
        k, v = plan.kv(0, p)
        ko, vo = o.kv(0, p)
        assert np.array_equal(k, ko) and np.array_equal(v, vo), p
    for p in range(1380, 1383):
        ref = o.forward(toks[p], p)
        assert np.array_equal(plan.tornadoVMForwardDecode(toks[p], p), ref), p
    plan.freeTornadoExecutionPlan()


def test_llama32_1b_shaped_layer_tied_vocab_decode_and_prefill512(pkg, orc, planmod):
    """BASELINE configs[1] (Llama-3.2-1B Q8_0 tg128) at ITS OWN shape: dim 2048 / hidden 8192 / head_size 64 and the tied
    128256 x 2048 head (wcls = token_embd): decode from position 0 with full logits, per-layer x and device argmax, a 512-token
    batched prefill (pp512 -b 512) and decode steps on top of it, all against the C oracle."""
    plan_mod, hip = planmod
    m = _model(pkg, "1b-layer", 109)
    assert m.cfg.tied and m.cfg.vocab == 128256 and m.cfg.hidden == 8192
    toks = pkg.javarand.bench_tokens(m.cfg.vocab, 516)
    plan = plan_mod.HipMasterPlan.initializeTornadoVMPlan(m, prefill_batch_size=512, flags=hip.FLAG_LAYER_TAPS)
    o = orc.COracle(m)
    for pos in range(4):
        ref, lx = o.forward(toks[pos], pos, layer_x=True)
        got = plan.tornadoVMForwardDecode(toks[pos], pos)
        assert got.shape == (128256,)
        assert np.array_equal(got, ref), pos
        assert np.array_equal(plan.layer_x(0), lx[0]), pos
        assert plan.forward_decode_argmax(toks[pos], pos) == orc.argmax(ref)
    plan.reset_kv()
    o2 = orc.COracle(m)
    plan.tornadoVMForwardBatchPrefill(toks[:512], 0)
    o2.prefill(toks[:512], 0)
    assert np.array_equal(plan.x(), o2.x())
    for p in (0, 127, 128, 511):
        k, v = plan.kv(0, p)
        ko, vo = o2.kv(0, p)
        assert np.array_equal(k, ko) and np.array_equal(v, vo), p
    for pos in range(512, 515):
        ref = o2.forward(toks[pos], pos)
        assert np.array_equal(plan.tornadoVMForwardDecode(toks[pos], pos), ref), pos
    plan.freeTornadoExecutionPlan()


@pytest.mark.parametrize("wtype", [2, 1])
def test_q4_0_and_f16_on_the_8b_layer_shape(pkg, orc, planmod, wtype):
    """BASELINE configs[3] / configs[0] dtypes at the 8B layer shape: Q4_0 and F16 matrices with K = 4096 and K = 14336 (the VL
    kernels' chunk loops at 16 / 56 chunks of 256 elements per row) in the reference's default Vector-API order; a 40-token prefill
    in two ragged chunks (the batched Vector-API-order prefill), then decode steps, against the oracle in the same mode."""
    plan_mod, hip = planmod
    m = _model(pkg, "8b-layer", 113, wtype=wtype)
    plan = plan_mod.HipMasterPlan.initializeTornadoVMPlan(m, prefill_batch_size=32, flags=hip.FLAG_LAYER_TAPS)
    o = orc.COracle(m, vector_bits=256)
    toks = pkg.javarand.bench_tokens(m.cfg.vocab, 46)
    for pos in range(3):
        ref, lx = o.forward(toks[pos], pos, layer_x=True)
        got = plan.tornadoVMForwardDecode(toks[pos], pos)
        assert np.array_equal(got, ref), pos
        assert np.array_equal(plan.layer_x(0), lx[0]), pos
        assert plan.forward_decode_argmax(toks[pos], pos) == orc.argmax(ref)
    plan.reset_kv()
    o2 = orc.COracle(m, vector_bits=256)
    plan.prefill(toks[:40], 0)                     # chunks of 32 + 8
    o2.prefill(toks[:40], 0)
    assert np.array_equal(plan.x(), o2.x())
    for p in (0, 31, 32, 39):
        k, v = plan.kv(0, p)
        ko, vo = o2.kv(0, p)
        assert np.array_equal(k, ko) and np.array_equal(v, vo), p
    for pos in range(40, 43):
        ref = o2.forward(toks[pos], pos)
        assert np.array_equal(plan.tornadoVMForwardDecode(toks[pos], pos), ref), pos
    plan.freeTornadoExecutionPlan()


def test_q4_0_tp8_on_the_8b_layer_shape(pkg, orc, planmod):
    """BASELINE configs[3] (Llama-3-8B Q4_0, TP = 8) at the 8B layer shape inside one GPU: eight row-split ranks (one kv head, 1792
    hidden units, 512 dim rows, 256 vocab rows each) as host threads over the peer-write gather; every rank's logits equal the
    single-GPU CPU oracle's.  (Ranks on eight DEVICES are the driver's SCALE run; nothing here crosses an xGMI link.)"""
    import threading
    plan_mod, hip = planmod
    tp = 8
    m = _model(pkg, "8b-layer", 127, wtype=2)
    o = orc.COracle(m, vector_bits=256)
    toks = pkg.javarand.bench_tokens(m.cfg.vocab, 2)
    ref = [o.forward(t, p) for p, t in enumerate(toks)]
    grp = plan_mod.make_local_group(tp)
    out, err = [None] * tp, [None] * tp

    def rank_main(r):
        try:
            plan = plan_mod.HipMasterPlan(m, tp_rank=r, tp_size=tp, local_group=grp)
            out[r] = [plan.forward_decode(t, p) for p, t in enumerate(toks)]
            plan.freeTornadoExecutionPlan()
        except Exception as e:   # noqa: BLE001
            err[r] = e

    th = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(tp)]
    [t.start() for t in th]
    [t.join(timeout=900) for t in th]
    assert all(e is None for e in err), err
    assert not any(t.is_alive() for t in th)
    hip.lib().gl3_local_group_destroy(grp)
    for r in range(tp):
        for p in range(len(toks)):
            assert np.array_equal(out[r][p], ref[p]), (r, p)


def test_vocab_128256_projection(pkg, orc, planmod):
    plan_mod, hip = planmod
    m = _model(pkg, "8b-vocab", 103)
    plan = plan_mod.HipMasterPlan(m)
    o = orc.COracle(m)
    for pos, t in enumerate([128000, 7, 128255]):            # begin-of-text id, a low id, the last row of the embedding
        ref = o.forward(t, pos)
        got = plan.tornadoVMForwardDecode(t, pos)
        assert got.shape == (128256,)
        assert np.array_equal(got, ref), pos
        assert plan.forward_decode_argmax(t, pos) == orc.argmax(ref)
    plan.freeTornadoExecutionPlan()


def test_static_batched_decode_b32_b33_b64_qwen3_4b_shape(pkg, orc, planmod):
    plan_mod, hip = planmod
    m = _model(pkg, "qwen3-4b-2l", 107)
    nseq = 64
    plan = plan_mod.HipMasterPlan(m, prefill_batch_size=64, n_seqs=nseq)
    oracles = [orc.COracle(m) for _ in range(nseq)]
    rng = np.random.default_rng(11)
    lens = [1 + (s % 5) for s in range(nseq)]
    cur, pos = [], []
    for s in range(nseq):
        prompt = rng.integers(0, m.cfg.vocab, lens[s]).tolist()
        plan.prefill_seq(s, prompt, 0)
        oracles[s].prefill(prompt, 0)
        cur.append(int(rng.integers(0, m.cfg.vocab)))
        pos.append(lens[s])

    def step(seqs):
        logits, ids = plan.forward_decode_batch([cur[s] for s in seqs], seqs, [pos[s] for s in seqs])
        for row, s in enumerate(seqs):
            ref = oracles[s].forward(cur[s], pos[s])
            assert np.array_equal(logits[row], ref), (len(seqs), row, s)
            assert int(ids[row]) == orc.argmax(ref)
            cur[s], pos[s] = int(ids[row]), pos[s] + 1          # greedy continuation per sequence

    step(list(range(32)))                    # B = 32: the configs[4] batch, every MFMA column live
    step(list(range(31, -1, -1)))            # same sequences, reversed batch rows
    step(list(range(20, 53)))                # B = 33: 32-token-tile GEMM, second tile has one live token
    step(list(range(nseq)))                  # B = 64: two full 32-token tiles
    step(list(range(5)))                     # and back to a small batch
    for s in (0, 31, 52, 63):
        k, v = plan.kv_seq(s, 1, pos[s] - 1)
        ko, vo = oracles[s].kv(1, pos[s] - 1)
        assert np.array_equal(k, ko) and np.array_equal(v, vo), s
    plan.freeTornadoExecutionPlan()
