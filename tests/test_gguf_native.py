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


YARN_CASES = [(64, 64, 1000000.0, 8.0, 32.0, 1.0, 1.0, 4096), (160, 128, 1000000.0, 48.0, 32.0, 1.0, 1.0, 8192),
              (33, 128, 1e9, 48.0, 32.0, 1.0, 0.0, 8192),          # log_multiplier 0: mscale = 1
              (20, 32, 10000.0, 4.0, 32.0, 1.0, 0.5, 2048), (9, 96, 500000.0, 2.5, 16.0, 2.0, 1.0, 1024)]


@pytest.mark.parametrize("ctx,hs,theta,factor,bf,bs,lm,oc", YARN_CASES)
def test_yarn_rope_table_four_statements_agree_bit_for_bit(pkg, hip, orc, ctx, hs, theta, factor, bf, bs, lm, oc):
    """RoPE.precomputeFreqsCisYaRN (RoPE.java:39-83): the library's table (what gl3_load_gguf uploads for a Devstral file), the C
    oracle's, and the two NumPy statements (host mirror, oracle_np) are the same bits."""
    from oracle import oracle_np
    cr, ci = pkg.synth.rope_table_yarn(ctx, hs, theta, factor, bf, bs, lm, oc)
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    a, b = np.empty_like(cr), np.empty_like(ci)
    hip.lib().gl3_rope_table_yarn(ctx, hs, C.c_float(theta), C.c_float(factor), C.c_float(bf), C.c_float(bs), C.c_float(lm), oc, p(a), p(b))
    assert np.array_equal(a, cr) and np.array_equal(b, ci)
    a2, b2 = np.empty_like(cr), np.empty_like(ci)
    orc.lib().orc_rope_table_yarn(ctx, hs, theta, factor, bf, bs, lm, oc, p(a2), p(b2))
    assert np.array_equal(a2, cr) and np.array_equal(b2, ci)
    n_cr, n_ci = oracle_np.rope_table_yarn(ctx, hs, theta, factor, bf, bs, lm, oc)
    assert np.array_equal(n_cr, cr) and np.array_equal(n_ci, ci)


def test_yarn_rope_table_known_structure(pkg):
    """Hand-checkable structure of the YaRN table (RoPE.java:46-69): below the fast-rotation correlation dimension the pair keeps the
    plain frequency, above the slow one it is divided by the factor, and every entry carries mscale = 1 + 0.1 * m * ln(factor)."""
    ctx, hs, theta, factor, bf, bs, lm, oc = 48, 128, 1000000.0, 48.0, 32.0, 1.0, 1.0, 8192
    cr, ci = (t.reshape(ctx, hs // 2) for t in pkg.synth.rope_table_yarn(ctx, hs, theta, factor, bf, bs, lm, oc))
    pr, pi = (t.reshape(ctx, hs // 2) for t in pkg.synth.rope_table(ctx, hs, theta))
    mscale = np.float32(1.0) + np.float32(0.1) * np.float32(np.log(np.float64(np.float32(48.0))))
    assert abs(float(mscale) - 1.3871201) < 1e-6
    # corr dims: 128 * ln(8192 / (32 * 2 pi)) / (2 ln 1e6) = 17.17..., 128 * ln(8192 / (2 pi)) / (2 ln 1e6) = 33.22...
    low, high = 17.17, 33.23
    fast = np.arange(hs // 2) <= int(low)              # ramp == 1: extrapolated (plain) frequency
    assert np.array_equal(cr[:, fast], (pr[:, fast] * mscale).astype(np.float32))
    assert np.array_equal(ci[:, fast], (pi[:, fast] * mscale).astype(np.float32))
    slow = np.arange(hs // 2) >= int(high) + 1         # ramp == 0: frequency / factor
    i = np.arange(0, hs, 2, dtype=np.float64)
    f = ((1.0 / np.power(theta, i / hs)).astype(np.float32) * (np.float32(1.0) / np.float32(factor))).astype(np.float32)
    val = (np.arange(ctx, dtype=np.float32)[:, None] * f[None, :]).astype(np.float32).astype(np.float64)
    assert np.array_equal(cr[:, slow], (np.cos(val).astype(np.float32) * mscale).astype(np.float32)[:, slow])
    mid = ~fast & ~slow
    assert mid.sum() == 16 and not np.array_equal(cr[1:, mid], (pr[1:, mid] * mscale).astype(np.float32))
    assert np.array_equal(cr[0], np.full(hs // 2, mscale, np.float32)) and not ci[0].any()     # position 0: cos = mscale, sin = 0


def test_devstral_gguf_metadata_round_trip(pkg, hip, tmp_path):
    """A "mistral3" file (DevstralModelLoader.java:45-110): Llama graph, head_size from attention.key_length, YaRN parameters."""
    m = pkg.synth.make_numpy(pkg.synth.CONFIGS["tiny-devstral"], seed=3)
    path = str(tmp_path / "d.gguf")
    m.write_gguf(path)
    L = hip.lib()
    g = C.c_void_p()
    hip.check_gguf(L.gl3_gguf_open(path.encode(), C.byref(g)))
    try:
        d = hip.ModelDesc()
        theta = C.c_float()
        hip.check_gguf(L.gl3_gguf_model_desc(g, C.byref(d), C.byref(theta)), g)
        c = m.cfg
        assert (d.arch, d.dim, d.n_heads, d.n_kv_heads, d.head_size) == (0, c.dim, c.n_heads, c.n_kv_heads, c.head_size)
        assert d.head_size * d.n_heads == 2 * d.dim
        f, bf, bs, lm, oc = C.c_float(), C.c_float(), C.c_float(), C.c_float(), C.c_int32()
        assert L.gl3_gguf_yarn_params(g, C.byref(f), C.byref(bf), C.byref(bs), C.byref(lm), C.byref(oc)) == 1
        assert (f.value, bf.value, bs.value, lm.value, oc.value) == c.yarn
    finally:
        L.gl3_gguf_close(g)
    back = pkg.synth.SynthModel.from_gguf(path)
    assert back.cfg.yarn == c.yarn and back.cfg.head_size == c.head_size and back.cfg.arch == 0
    assert np.array_equal(back.rope[0], m.rope[0]) and np.array_equal(back.rope[1], m.rope[1])
    # a plain file has no YaRN block
    m2 = pkg.synth.make_numpy(pkg.synth.CONFIGS["tiny-llama"], seed=3)
    m2.write_gguf(path)
    hip.check_gguf(L.gl3_gguf_open(path.encode(), C.byref(g)))
    try:
        assert L.gl3_gguf_yarn_params(g, None, None, None, None, None) == 0
    finally:
        L.gl3_gguf_close(g)


def _write_with_extra_metadata(model, path, extra):
    """model.write_gguf with extra metadata keys (the synthetic writer only emits rope.scaling.* for cfg.yarn models)."""
    base = model.metadata
    model.metadata = lambda: {**base(), **extra}
    try:
        model.write_gguf(path)
    finally:
        model.metadata = base


@pytest.mark.parametrize("cfg", ["tiny-qwen3", "tiny-llama"])
def test_yarn_keys_outside_mistral3_are_ignored(pkg, hip, tmp_path, cfg):
    """Only DevstralModelLoader reads <arch>.rope.scaling.* (DevstralModelLoader.java:80-93); LlamaModelLoader.java:68 and
    Qwen3ModelLoader.java:78 call RoPE.precomputeFreqsCis whatever the file says.  A qwen3 / llama file that carries yarn keys
    (128K variants do) must therefore get the PLAIN table in the native loader and in the Python reader (r4 advisor finding)."""
    m = pkg.synth.make_numpy(pkg.synth.CONFIGS[cfg], seed=3)
    a = "qwen3" if "qwen3" in cfg else "llama"
    path = str(tmp_path / "y.gguf")
    _write_with_extra_metadata(m, path, {f"{a}.rope.scaling.type": "yarn", f"{a}.rope.scaling.factor": 4.0, f"{a}.rope.scaling.yarn_beta_fast": 32.0,
                                         f"{a}.rope.scaling.yarn_beta_slow": 1.0, f"{a}.rope.scaling.original_context_length": 64})
    L = hip.lib()
    g = C.c_void_p()
    hip.check_gguf(L.gl3_gguf_open(path.encode(), C.byref(g)))
    try:
        assert L.gl3_gguf_yarn_params(g, None, None, None, None, None) == 0
    finally:
        L.gl3_gguf_close(g)
    back = pkg.synth.SynthModel.from_gguf(path)
    assert back.cfg.yarn is None
    assert np.array_equal(back.rope[0], m.rope[0]) and np.array_equal(back.rope[1], m.rope[1])


@pytest.mark.parametrize("key,val", [("factor", 0.0), ("factor", -2.0), ("original_context_length", 0)])
def test_unusable_yarn_parameters_are_rejected(pkg, hip, tmp_path, key, val):
    """factor == 0 would put 1 / 0 into every interpolated frequency (NaN table); the native reader reports -1 and gl3_load_gguf /
    the Python reader refuse the file."""
    m = pkg.synth.make_numpy(pkg.synth.CONFIGS["tiny-devstral"], seed=3)
    path = str(tmp_path / "bad.gguf")
    _write_with_extra_metadata(m, path, {f"mistral3.rope.scaling.{key}": val})
    L = hip.lib()
    g = C.c_void_p()
    hip.check_gguf(L.gl3_gguf_open(path.encode(), C.byref(g)))
    try:
        assert L.gl3_gguf_yarn_params(g, None, None, None, None, None) == -1
    finally:
        L.gl3_gguf_close(g)
    with pytest.raises(ValueError):
        pkg.synth.SynthModel.from_gguf(path)
    ctx = C.c_void_p()
    opts = hip.ModelDesc()
    opts.struct_size = C.sizeof(hip.ModelDesc)
    opts.ctx = 64
    assert L.gl3_load_gguf(path.encode(), C.byref(opts), C.byref(ctx)) == hip.E_ARG      # refused before any device work
    assert b"rope.scaling" in L.gl3_gguf_last_error(None)


def test_qwen2moe_gguf_metadata_round_trip(pkg, hip, tmp_path):
    """A "qwen2moe" file (Qwen2MoEModelLoader.java:56-110): expert counts from the metadata, the experts' hidden size from the first
    dimension of the 3-D blk.0.ffn_down_exps.weight, the shared expert's from feed_forward_length; stacked experts are 3-D tensors."""
    m = pkg.synth.make_numpy(pkg.synth.CONFIGS["tiny-qwen2moe"], seed=3)
    path = str(tmp_path / "moe.gguf")
    m.write_gguf(path)
    L = hip.lib()
    g = C.c_void_p()
    hip.check_gguf(L.gl3_gguf_open(path.encode(), C.byref(g)))
    try:
        d = hip.ModelDesc()
        hip.check_gguf(L.gl3_gguf_model_desc(g, C.byref(d), None), g)
        c = m.cfg
        assert (d.arch, d.dim, d.hidden, d.n_experts, d.n_experts_used, d.moe_hidden) == (5, c.dim, c.hidden, c.n_experts, c.n_experts_used, c.moe_hidden)
        dims = {}
        for i in range(L.gl3_gguf_tensor_count(g)):
            name, ty, ne = C.c_char_p(), C.c_int32(), (C.c_uint64 * 4)()
            assert L.gl3_gguf_tensor_info(g, i, C.byref(name), C.byref(ty), ne, None, None) == 0
            dims[name.value.decode()] = (ty.value, list(ne))
        assert dims["blk.0.ffn_gate_exps.weight"] == (8, [c.dim, c.moe_hidden, c.n_experts, 1])
        assert dims["blk.1.ffn_down_exps.weight"] == (8, [c.moe_hidden, c.dim, c.n_experts, 1])
        assert dims["blk.0.ffn_gate_inp.weight"] == (0, [c.dim, c.n_experts, 1, 1])
        assert dims["blk.0.ffn_gate_inp_shexp.weight"][0] == 0
    finally:
        L.gl3_gguf_close(g)
    back = pkg.synth.SynthModel.from_gguf(path)
    assert (back.cfg.arch, back.cfg.n_experts, back.cfg.n_experts_used, back.cfg.moe_hidden, back.cfg.hidden) == (5, c.n_experts, c.n_experts_used, c.moe_hidden, c.hidden)
    for name, (raw, ty, rows, cols) in m.tensors.items():
        r2, t2, rows2, cols2 = back.tensors[name]
        assert (t2, rows2, cols2) == (ty, rows, cols) and np.array_equal(np.asarray(r2).view(np.uint8).reshape(-1), raw.view(np.uint8).reshape(-1)), name


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
