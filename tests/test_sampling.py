"""Sampling: the oracle's restatement of Sampler.selectSampler + CategoricalSampler / ToppSampler against an
independent NumPy / pure-Python restatement on small vocabularies, hand-checkable cases, and the Python twin of the JDK's default
RandomGenerator (structure only: it is UNPINNED, see javarand.py)."""
import numpy as np
import pytest

F32 = np.float32


def np_softmax_t(logits, temperature):
    """divideInPlace(temperature) + FloatTensor.softmaxInPlace (J/tensor/standard/FloatTensor.java:203-219) in NumPy."""
    p = (np.asarray(logits, F32) / F32(temperature)).astype(F32)
    m = np.max(p)
    e = np.exp((p - m).astype(np.float64)).astype(F32)
    s = np.take(np.add.accumulate(e, dtype=F32), -1)
    return (e / s).astype(F32)


def py_categorical(p, coin):
    cdf = F32(0)
    for i, v in enumerate(p):
        cdf = F32(cdf + v)
        if F32(coin) < cdf:
            return i
    return len(p) - 1


def py_topp(p, topp, coin):
    """ToppSampler.sampleFromFloatTensor + processTopP (J/inference/sampler/ToppSampler.java:57-160), written from the Java."""
    n = len(p)
    idx = [0] * n
    head, tail = 0, n - 1
    cutoff = F32(F32(1.0 - F32(topp)) / F32(n - 1))
    for i in range(n):
        if p[i] >= cutoff:
            idx[head] = i; head += 1
        else:
            idx[tail] = i; tail -= 1
    n0 = head

    def cmp(a, b):
        return -1 if float(p[b]) < float(p[a]) else (1 if float(p[b]) > float(p[a]) else 0)

    def sift(frm, m):
        prev = frm
        while 2 * prev + 1 < m:
            nxt = 2 * prev + 1
            r = 2 * prev + 2
            if r < m and cmp(idx[r], idx[nxt]) < 0:
                nxt = r
            if cmp(idx[nxt], idx[prev]) < 0:
                idx[prev], idx[nxt] = idx[nxt], idx[prev]
                prev = nxt
            else:
                break

    for i in range(n0 // 2 - 1, -1, -1):
        sift(i, n0)
    cum, last = F32(0), 0
    for i in range(n0 - 1, -1, -1):
        idx[0], idx[i] = idx[i], idx[0]
        cum = F32(cum + p[idx[i]])
        if cum > F32(topp):
            last = i
            break
        sift(0, i - 1)
    r = F32(F32(coin) * cum)
    cdf = F32(0)
    for i in range(n0 - 1, last - 1, -1):
        cdf = F32(cdf + p[idx[i]])
        if r < cdf:
            return idx[i]
    return idx[last]


def test_oracle_sampler_matches_independent_restatement(orc):
    rng = np.random.default_rng(3)
    for n in (5, 64, 517):
        for trial in range(12):
            logits = (rng.standard_normal(n) * rng.choice([0.3, 2.0, 8.0])).astype(F32)
            t = float(rng.choice([0.5, 0.8, 1.0, 1.7]))
            coin = float(F32(rng.random()))
            p = np_softmax_t(logits, t)
            tok, probs = orc.sample(logits, t, 0.0, coin, want_probs=True)
            assert np.array_equal(probs, p)
            assert tok == py_categorical(p, coin)
            for topp in (0.5, 0.9, 0.95):
                assert orc.sample(logits, t, topp, coin) == py_topp(p, topp, coin), (n, trial, topp)
    # temperature 0 = greedy argmax (first index of the maximum); topp outside (0, 1) = categorical
    lg = np.array([0.1, 3.0, 3.0, -1.0], F32)
    assert orc.sample(lg, 0.0, 0.9, 0.5) == 1
    assert orc.sample(lg, 1.0, 1.0, 0.0) == orc.sample(lg, 1.0, 0.0, 0.0) == 0


def test_hand_checked_cases(orc):
    # two equal logits: p = [0.5, 0.5]; coin < 0.5 -> 0, else 1; coin = 0.5 is NOT < cdf_0 = 0.5 -> 1
    lg = np.array([1.0, 1.0], F32)
    assert orc.sample(lg, 1.0, 0.0, 0.49) == 0 and orc.sample(lg, 1.0, 0.0, 0.5) == 1
    # top-p 0.6 on p = [0.7, 0.2, 0.1]: the largest alone exceeds topp -> always index of the 0.7
    lg = np.log(np.array([0.7, 0.2, 0.1], np.float64)).astype(F32)
    for coin in (0.0, 0.3, 0.99):
        assert orc.sample(lg, 1.0, 0.6, coin) == 0
    # top-p 0.8: {0.7, 0.2} kept, r = coin * 0.9: coin 0.7 -> r 0.63 < 0.7 -> 0; coin 0.8 -> r 0.72 -> 1
    assert orc.sample(lg, 1.0, 0.8, 0.7) == 0 and orc.sample(lg, 1.0, 0.8, 0.8) == 1


def test_default_random_generator_twin_is_a_well_formed_l32x64(pkg):
    """Structure of the UNPINNED twin of RandomGeneratorFactory.getDefault(): deterministic, seed-sensitive, floats in [0, 1) on
    a 2^-24 grid, and the LCG / xoroshiro state updates are the published ones (period checks on a short horizon)."""
    R = pkg.javarand.L32X64MixRandom
    a, b, c = R(1234), R(1234), R(1235)
    xs = [a.next_float() for _ in range(2000)]
    assert xs == [b.next_float() for _ in range(2000)] and xs[:8] != [c.next_float() for _ in range(8)]
    assert all(0.0 <= x < 1.0 and (x * (1 << 24)).is_integer() for x in xs)
    assert 0.45 < sum(xs) / len(xs) < 0.55 and len(set(xs)) > 1990
    g = R(7)
    assert g.a & 1 == 1 and g.s == 1
    s0 = g.s
    g.next_int()
    assert g.s == (0xADB4A92D * s0 + g.a) & 0xFFFFFFFF
