"""The exact parallel evaluation of the sequential f32 sum of squares (csrc/gl3_seqsum.h) against the plain
left-to-right chain, on random and adversarial inputs (ties, binade crossings, zeros, outliers)."""
import ctypes as C

import numpy as np
import pytest

import __graft_entry__ as ge

pytestmark = pytest.mark.gpu


def seq_sumsq(x):
    a = (x.astype(np.float32) * x.astype(np.float32)).astype(np.float32)
    return np.add.accumulate(a, dtype=np.float32)[-1]


def cases():
    rng = np.random.default_rng(5)
    for n in (1024, 2048, 2560, 4096, 5120):
        yield "normal", (rng.standard_normal(n) * 0.02).astype(np.float32)
        yield "uniform", (rng.random(n) * 2 - 1).astype(np.float32)
        yield "small-ints", rng.integers(0, 8, n).astype(np.float32)               # exact adds and ties
        yield "pow2", np.ldexp(1.0, rng.integers(-6, 6, n)).astype(np.float32)
        yield "const", np.ones(n, np.float32)
        yield "quarters", (rng.integers(1, 5, n) * 0.25).astype(np.float32)        # many exact ties
        x = (rng.random(n) * 1e-3).astype(np.float32); x[rng.integers(0, n, n // 50)] = 100.0
        yield "outliers", x
        x = rng.random(n).astype(np.float32); x[:5] = 1e-12
        yield "tiny-head", x
        yield "wide", np.ldexp(rng.random(n), rng.integers(-20, 20, n)).astype(np.float32)
        x = (rng.random(n) * 0.01).astype(np.float32); x[n // 2] = 3000.0
        yield "one-huge", x
        yield "zeros", np.zeros(n, np.float32)


def test_exact_sumsq_matches_sequential_chain(pkg):
    from importlib import import_module
    hip = import_module(ge.PKG_NAME + ".hip")
    L = hip.lib()
    out = C.c_float()
    for name, x in cases():
        x = np.ascontiguousarray(x, np.float32)
        assert L.gl3_debug_sumsq(0, x.ctypes.data_as(C.c_void_p), x.size, C.byref(out)) == 0
        ref = seq_sumsq(x)
        assert np.float32(out.value).view(np.uint32) == np.float32(ref).view(np.uint32), (name, x.size, out.value, float(ref))
