"""Native GGUF reader (gl3_gguf_*): header / metadata / tensor table parsed in C++ against the Python writer + reader
(gpullama3.java_amd/gguf.py), config extraction as LlamaModelLoader.java:47-69 / Qwen3ModelLoader.java:48-79, the native
RoPE table against the NumPy restatement of RoPE.precomputeFreqsCis.  Host-only: no GPU call."""
import ctypes as C
import os

import numpy as np
import pytest

import __graft_entry__ as ge


@pytest.fixture(scope="module")
def hip(pkg):
    from importlib import import_module
    return import_module(ge.PKG_NAME + ".hip")


@pytest.mark.parametrize("cfg,wtype", [("tiny-llama", 8), ("tiny-qwen3", 8), ("tiny-llama-tied", 2), ("tiny-llama", 1), ("tiny-qwen2", 8)])
def test_native_reader_agrees_with_the_python_reader(pkg, hip, tmp_path, cfg, wtype):
    m = pkg.synth.make_numpy(pkg.synth.CONFIGS[cfg], wtype=wtype, seed=3)
    path = str(tmp_path / "m.gguf")
    m.write_gguf(path)
    L = hip.lib()
    g = C.c_void_p()
    hip.check_gguf(L.gl3_gguf_open(path.encode(), C.byref(g)))
    try:
        assert L.gl3_gguf_tensor_count(g) == len(m.tensors)
        seen = {}
        for i in range(L.gl3_gguf_tensor_count(g)):
            name, ty, ne, data, nbytes = C.c_char_p(), C.c_int32(), (C.c_uint64 * 4)(), C.c_void_p(), C.c_uint64()
            assert L.gl3_gguf_tensor_info(g, i, C.byref(name), C.byref(ty), ne, C.byref(data), C.byref(nbytes)) == 0
            seen[name.value.decode()] = (ty.value, list(ne), C.string_at(data.value, nbytes.value))
        for name, (raw, ty, rows, cols) in m.tensors.items():
            t = seen[name]
            assert t[0] == ty and t[1][0] == cols and (rows == 1 or t[1][1] == rows)
            assert t[2] == raw.tobytes(), name
        d = hip.ModelDesc()
        theta = C.c_float()
        hip.check_gguf(L.gl3_gguf_model_desc(g, C.byref(d), C.byref(theta)), g)
        c = m.cfg
        assert (d.arch, d.dim, d.hidden, d.n_layers, d.n_heads, d.n_kv_heads, d.head_size, d.vocab, d.ctx, d.weight_type) == \
            (c.arch, c.dim, c.hidden, c.n_layers, c.n_heads, c.n_kv_heads, c.head_size, c.vocab, c.ctx, wtype)
        assert np.float32(d.rms_eps) == np.float32(c.rms_eps) and np.float32(theta.value) == np.float32(c.rope_theta)
        d2 = hip.ModelDesc()
        d2.ctx = 16                                   # caller may ask for a shorter KV cache
        hip.check_gguf(L.gl3_gguf_model_desc(g, C.byref(d2), None), g)
        assert d2.ctx == 16
        v = C.c_double()
        a = {0: "llama", 1: "qwen3", 2: "qwen2"}[c.arch]
        assert L.gl3_gguf_meta_number(g, (a + ".block_count").encode(), C.byref(v)) == 0 and v.value == c.n_layers
        s = C.c_char_p()
        assert L.gl3_gguf_meta_string(g, b"general.architecture", C.byref(s)) == 0 and s.value.decode() == a
        assert L.gl3_gguf_meta_number(g, b"no.such.key", C.byref(v)) != 0
    finally:
        L.gl3_gguf_close(g)


@pytest.mark.parametrize("ctx,hs,theta", [(64, 32, 500000.0), (40, 128, 1000000.0), (512, 64, 10000.0)])
def test_native_rope_table_is_bit_identical_to_the_numpy_restatement(pkg, hip, ctx, hs, theta):
    cr, ci = pkg.synth.rope_table(ctx, hs, theta)
    a, b = np.empty_like(cr), np.empty_like(ci)
    hip.lib().gl3_rope_table(ctx, hs, C.c_float(theta), a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p))
    assert np.array_equal(a, cr) and np.array_equal(b, ci)


def test_reader_rejects_bad_files(hip, tmp_path):
    L = hip.lib()
    g = C.c_void_p()
    p = tmp_path / "bad.gguf"
    p.write_bytes(b"GGML" + b"\0" * 64)
    assert L.gl3_gguf_open(str(p).encode(), C.byref(g)) == -1 and b"magic" in L.gl3_gguf_last_error(None)
    p.write_bytes(b"GGUF" + (3).to_bytes(4, "little") + (1 << 40).to_bytes(8, "little") + (0).to_bytes(8, "little"))
    assert L.gl3_gguf_open(str(p).encode(), C.byref(g)) == -1
    p.write_bytes(b"GGUF" + (1).to_bytes(4, "little") + b"\0" * 16)
    assert L.gl3_gguf_open(str(p).encode(), C.byref(g)) == -2        # GGUF v1: unsupported
    assert L.gl3_gguf_open(b"/nonexistent/file.gguf", C.byref(g)) == -1
