"""gl3_forward_decode_sample on the GPU against the oracle's Sampler.selectSampler restatement: same probabilities (bit for bit,
including the 128 k-long strictly sequential softmax sum evaluated in parallel chunks) and the same sampled id for the same
rng.nextFloat(1f), for categorical and top-p sampling, small and full-size vocabularies."""
import numpy as np
import pytest

import __graft_entry__ as ge

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def planmod():
    from importlib import import_module
    ge.load_package()
    return import_module(ge.PKG_NAME + ".plan"), import_module(ge.PKG_NAME + ".hip")


@pytest.mark.parametrize("cfg", ["tiny-llama", "mid-llama"])
def test_sampled_decode_matches_the_oracle_sampler(pkg, orc, planmod, cfg):
    plan_mod, hip = planmod
    m = pkg.synth.make_numpy(pkg.synth.CONFIGS[cfg], seed=61)
    plan = plan_mod.HipMasterPlan(m)
    o = orc.COracle(m)
    rng = pkg.javarand.L32X64MixRandom(1234)            # the reference's default seed (Options / OpenAIServer.java:252)
    tok = 1
    for pos in range(10):
        temperature, topp = [(0.7, 0.95), (1.0, 0.0), (0.4, 0.5), (1.3, 1.0), (0.0, 0.9)][pos % 5]
        coin = rng.next_float()
        logits = o.forward(tok, pos)
        want, probs = orc.sample(logits, temperature, topp, coin, want_probs=True) if temperature > 0 else (orc.argmax(logits), None)
        got = plan.forward_decode_sample(tok, pos, temperature, topp, coin)
        if temperature > 0:
            assert np.array_equal(plan.sample_probs(), probs), pos
        assert got == want, (pos, temperature, topp, coin)
        tok = got
    with pytest.raises(hip.Gl3Error):
        plan.forward_decode_sample(1, 10, 0.7, 0.9, 1.0)          # coin must be < 1
    plan.freeTornadoExecutionPlan()


def test_full_vocabulary_softmax_and_sampling(pkg, orc, planmod):
    """vocab 128256 (31 exact chunks of 4096 + one of 1280): peaked and nearly flat distributions, coins that land in the first,
    a middle and the last chunk."""
    plan_mod, hip = planmod
    import torch
    cfg = pkg.synth.CONFIGS["8b-vocab"]
    m = pkg.synth.make_torch(cfg, seed=67, device="cuda" if torch.cuda.is_available() else "cpu")
    plan = plan_mod.HipMasterPlan(m)
    o = orc.COracle(m)
    logits = o.forward(128000, 0)
    for temperature, topp, coin in [(1.0, 0.0, 0.0), (1.0, 0.0, 0.5), (1.0, 0.0, 0.999999), (0.05, 0.0, 0.37), (0.7, 0.95, 0.61),
                                    (0.02, 0.9, 0.2), (5.0, 0.0, 0.77)]:
        want, probs = orc.sample(logits, temperature, topp, coin, want_probs=True)
        got = plan.forward_decode_sample(128000, 0, temperature, topp, coin)
        assert np.array_equal(plan.sample_probs(), probs), (temperature, topp)
        assert got == want, (temperature, topp, coin)
    plan.freeTornadoExecutionPlan()


def test_top_p_runs_on_the_device_and_ties_fall_back_to_the_heap(pkg, orc, planmod):
    """r5: ToppSampler on the device (radix sort of the candidates + exact chunked prefix sums, gl3_sample.hip) — 8 bytes come back instead
    of vocab x 4.  200 draws over the 128256-entry vocabulary at peaked, medium and nearly flat temperatures: every id equals the oracle's
    heap selection; most draws are answered on the device, the rest (a tie between equal probabilities at the sampled rank, where only the
    reference's heap history names the token) by the host heap — both counted by gl3_get_topp_counts."""
    plan_mod, hip = planmod
    import torch
    cfg = pkg.synth.CONFIGS["8b-vocab"]
    m = pkg.synth.make_torch(cfg, seed=71, device="cuda" if torch.cuda.is_available() else "cpu")
    plan = plan_mod.HipMasterPlan(m)
    o = orc.COracle(m)
    logits = o.forward(128000, 0)
    rng = pkg.javarand.L32X64MixRandom(99)
    n = 0
    for temperature in (0.02, 0.1, 0.7, 1.0, 3.0):
        for topp in (0.5, 0.9, 0.95, 0.999):
            for _ in range(10):
                coin = rng.next_float()
                want = orc.sample(logits, temperature, topp, coin)
                got = plan.forward_decode_sample(128000, 0, temperature, topp, coin)
                assert got == want, (temperature, topp, coin)
                n += 1
    dev, host = plan.topp_counts()
    assert dev + host == n and dev >= 0.8 * n, (dev, host)
    plan.freeTornadoExecutionPlan()
