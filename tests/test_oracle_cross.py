"""The C oracle (timed CPU baseline) against the committed golden fixtures (made by the independent
NumPy restatement) — bit for bit — plus structural properties of the reference path."""
import os

import numpy as np
import pytest

from oracle import oracle_np

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = [("tiny_llama_q8_0", "tiny-llama", 8, 7, 0), ("tiny_llama_f16", "tiny-llama", 1, 7, 0),
         ("tiny_llama_tied_q4_0", "tiny-llama-tied", 2, 11, 0), ("tiny_qwen3_q8_0", "tiny-qwen3", 8, 5, 0),
         ("tiny_qwen2_q8_0", "tiny-qwen2", 8, 13, 0), ("tiny_granite_q8_0", "tiny-granite", 8, 19, 0), ("tiny_phi3_q8_0", "tiny-phi3", 8, 23, 0),
         ("tiny_devstral_q8_0", "tiny-devstral", 8, 29, 0),
         # Vector-API dot order (256-bit species) for F16 / Q4_0 matrices: FP16FloatTensor.vectorDot / Q4_0FloatTensor.vectorDot
         ("tiny_llama_f16_v256", "tiny-llama", 1, 7, 256), ("tiny_llama_tied_q4_0_v256", "tiny-llama-tied", 2, 11, 256),
         # Q8_0 + 256 = -Dllama.quantizeActivation=false: Q8_0FloatTensor.vectorDot on the f32 activation (SURVEY 8 a4')
         ("tiny_llama_q8_0_f32act_v256", "tiny-llama", 8, 7, 256)]


@pytest.mark.parametrize("fx,cfg,wt,seed,vbits", CASES)
def test_c_oracle_matches_golden_bitwise(pkg, orc, fx, cfg, wt, seed, vbits):
    g = np.load(os.path.join(GOLD, fx + ".npz"))
    m = pkg.synth.make_numpy(pkg.synth.CONFIGS[cfg], wtype=wt, seed=seed)
    o = orc.COracle(m, vector_bits=vbits, f32_activation=(wt == 8 and vbits == 256))
    toks = g["tokens"]
    n_prompt = len(g["prompt"])
    assert toks[:n_prompt].tolist() == pkg.javarand.bench_tokens(m.cfg.vocab, n_prompt)
    for pos in range(g["logits"].shape[0]):
        lg, lx = o.forward(int(toks[pos]), pos, layer_x=True)
        assert np.array_equal(lg.view(np.uint32), g["logits"][pos].view(np.uint32)), pos
        if pos >= n_prompt - 1:
            assert orc.argmax(lg) == toks[pos + 1]            # greedy ids
    assert np.array_equal(lx, g["last_layer_x"])
    for l in range(m.cfg.n_layers):
        k, v = o.kv(l, g["logits"].shape[0] - 1)
        assert np.array_equal(k, g["k_last"][l]) and np.array_equal(v, g["v_last"][l])


def test_numpy_and_c_agree_on_fresh_seed(pkg, orc):
    # qwen2: q/k/v bias + NeoX RoPE; mha-llama: n_heads == n_kv_heads (kvMul = 1), head_size 128
    for cfg, wt in [("tiny-llama-tied", 8), ("tiny-qwen3", 1), ("tiny-qwen2", 8), ("tiny-qwen2", 2), ("mha-llama", 8), ("tiny-granite", 8), ("tiny-granite", 1), ("tiny-phi3", 8), ("tiny-phi3", 2),
                    ("tiny-devstral", 8), ("tiny-devstral", 1)]:       # devstral: q_dim 512 on dim 256, YaRN table
        m = pkg.synth.make_numpy(pkg.synth.CONFIGS[cfg], wtype=wt, seed=1234)
        co = orc.COracle(m)
        no = oracle_np.NpOracle(m.oracle_cfg(), m.oracle_tensors(), m.rope)
        for pos, t in enumerate(pkg.javarand.bench_tokens(m.cfg.vocab, 4, seed=9)):
            assert np.array_equal(co.forward(t, pos), no.forward(t, pos))


def test_vector_api_dot_order_numpy_and_c_agree_and_stay_close_to_scalar(pkg, orc):
    """The 256-bit Vector-API dots (8 fused accumulator lanes, DAZ f16 conversion, lane-order reduce) in both restatements,
    on a seed that is not a fixture; the scalar and vector orders of the reference agree to ~1e-3 (F16: DAZ flushes the
    ~0.2 % subnormal weights) / ~1e-6 (Q4_0) relative, with identical greedy ids on these models."""
    for cfg, wt, tol in [("tiny-llama", 1, 2e-3), ("tiny-qwen3", 2, 1e-4), ("tiny-llama-tied", 2, 1e-4)]:
        m = pkg.synth.make_numpy(pkg.synth.CONFIGS[cfg], wtype=wt, seed=4321)
        cv, cs = orc.COracle(m, vector_bits=256), orc.COracle(m)
        nv = oracle_np.NpOracle(m.oracle_cfg(), m.oracle_tensors(), m.rope, vector_bits=256)
        for pos, t in enumerate(pkg.javarand.bench_tokens(m.cfg.vocab, 4, seed=9)):
            a, b, c = cv.forward(t, pos), nv.forward(t, pos), cs.forward(t, pos)
            assert np.array_equal(a, b), (cfg, pos)
            assert not np.array_equal(a, c)                                    # a different rounding order, visibly
            assert float(np.max(np.abs(a - c)) / np.max(np.abs(c))) < tol
            assert orc.argmax(a) == orc.argmax(c)


def test_q8_0_f32_activation_vector_dot_numpy_and_c_agree(pkg, orc):
    """SURVEY 8 a4': Q8_0FloatTensor.vectorDot (llama.quantizeActivation=false) in both restatements on a fresh seed, and against
    the default int8-activation path: different arithmetic (no activation rounding), so logits differ at the 1e-2 level while the
    greedy ids of these models agree."""
    for cfg in ("tiny-llama", "tiny-qwen3", "tiny-granite"):
        m = pkg.synth.make_numpy(pkg.synth.CONFIGS[cfg], wtype=8, seed=2468)
        cv, cq = orc.COracle(m, vector_bits=256, f32_activation=True), orc.COracle(m)
        nv = oracle_np.NpOracle(m.oracle_cfg(), m.oracle_tensors(), m.rope, vector_bits=256, f32_activation=True)
        for pos, t in enumerate(pkg.javarand.bench_tokens(m.cfg.vocab, 4, seed=9)):
            a, b, c = cv.forward(t, pos), nv.forward(t, pos), cq.forward(t, pos)
            assert np.array_equal(a, b), (cfg, pos)
            assert not np.array_equal(a, c)
            assert float(np.max(np.abs(a - c)) / np.max(np.abs(c))) < 5e-2


def test_fma32_emulation_is_correctly_rounded():
    """oracle_np.fma32 (float64 product + round-to-odd sum) against exact rational arithmetic."""
    from fractions import Fraction
    rng = np.random.default_rng(5)
    a = rng.standard_normal(3000).astype(np.float32)
    b = rng.standard_normal(3000).astype(np.float32)
    c = (rng.standard_normal(3000) * rng.choice([1e-7, 1e-3, 1.0, 1e4], 3000)).astype(np.float32)
    got = oracle_np.fma32(a, b, c)
    for i in range(3000):
        exact = Fraction(float(a[i])) * Fraction(float(b[i])) + Fraction(float(c[i]))
        f = np.float32(float(exact))
        cands = [np.nextafter(f, np.float32(-np.inf)), f, np.nextafter(f, np.float32(np.inf))]
        best = min(cands, key=lambda z: (abs(Fraction(float(z)) - exact), int(np.float32(z).view(np.uint32)) & 1))
        assert best == got[i], i


def test_prefill_writes_the_same_kv_as_sequential_decode(pkg, orc):
    # InferenceCoreBatchPrefillDecode.batchForwardJavaPrefill :62-168 == forwardJava minus logits
    m = pkg.synth.make_numpy(pkg.synth.CONFIGS["tiny-llama"], seed=7)
    toks = pkg.javarand.bench_tokens(m.cfg.vocab, 9)
    a, b = orc.COracle(m), orc.COracle(m)
    for pos, t in enumerate(toks[:8]):
        a.forward(t, pos)
    b.prefill(toks[:5], 0)
    b.prefill(toks[5:8], 5)
    for l in range(m.cfg.n_layers):
        for pos in range(8):
            ka, va = a.kv(l, pos)
            kb, vb = b.kv(l, pos)
            assert np.array_equal(ka, kb) and np.array_equal(va, vb)
    assert np.array_equal(a.forward(toks[8], 8), b.forward(toks[8], 8))


def test_thread_count_independent(pkg, orc):
    import subprocess, sys, json
    code = ("import sys,json,numpy as np;sys.path.insert(0,%r);import __graft_entry__ as ge;p=ge.load_package();"
            "from oracle import oracle_c as oc;m=p.synth.make_numpy(p.synth.CONFIGS['tiny-llama'],seed=7);o=oc.COracle(m);"
            "o.forward(5,0);print(json.dumps(o.forward(9,1).view(np.uint32)[:64].tolist()))") % os.path.dirname(GOLD[:-6])
    outs = []
    for n in ("1", "3"):
        env = dict(os.environ, OMP_NUM_THREADS=n)
        outs.append(subprocess.check_output([sys.executable, "-c", code], env=env).decode().strip().splitlines()[-1])
    assert outs[0] == outs[1]
