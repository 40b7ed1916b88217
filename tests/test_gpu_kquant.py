"""A K-quant GGUF (Q4_K / Q5_K / Q6_K matrices mixed as in a Q4_K_M file) through the native loader: every matrix is converted to
Q8_0 at load (ModelLoader.dequantizeToQ8_0TornadoTensor semantics) and the plan then decodes bit-identically to the CPU oracle
running on the NumPy-converted Q8_0 tensors."""
import numpy as np
import pytest

import __graft_entry__ as ge
import kquant_np as kq

pytestmark = pytest.mark.gpu


def test_kquant_gguf_loads_as_q8_0_and_matches_the_oracle(pkg, orc, tmp_path):
    from importlib import import_module
    plan_mod = import_module(ge.PKG_NAME + ".plan")
    synth = pkg.synth
    cfg = synth.CONFIGS["tiny-llama"]
    rng = np.random.default_rng(77)
    base = synth.make_numpy(cfg, seed=7)                       # norm weights (F32) come from here
    kt, q8 = {}, {}
    for name, (raw, ty, rows, cols) in base.tensors.items():
        if ty != synth.GGML_Q8_0:
            kt[name] = q8[name] = (raw, ty, rows, cols)
            continue
        t = 14 if ("attn_v" in name or "ffn_down" in name or name == "output.weight") else 13 if "attn_k" in name else 12
        kraw = kq.random_blocks(t, rows * cols, rng, scale=4e-3 if t == 14 else 1.5e-3)
        kt[name] = (kraw, t, rows, cols)
        q8[name] = (kq.to_q8_0(kq.DEQUANT[t](kraw, rows * cols)), synth.GGML_Q8_0, rows, cols)
    km = synth.SynthModel(cfg, synth.GGML_Q8_0, kt)
    path = str(tmp_path / "kq.gguf")
    km.write_gguf(path)
    plan = plan_mod.HipMasterPlan.from_gguf(path, prefill_batch_size=8)
    o = orc.COracle(synth.SynthModel(cfg, synth.GGML_Q8_0, q8))
    toks = pkg.javarand.bench_tokens(cfg.vocab, 10)
    plan.prefill(toks[:4], 0)
    o.prefill(toks[:4], 0)
    for pos in range(4, 10):
        assert np.array_equal(plan.forward_decode(toks[pos], pos), o.forward(toks[pos], pos)), pos
    plan.freeTornadoExecutionPlan()
