"""Pins the oracle (and through it the HIP path) to dumps of the REAL reference, when a maintainer has produced them.

The reference holds no golden vectors for the forward pass and its JVM cannot run in this image, so parity is
UNPINNED until someone with a JDK runs the two commands of INTEGRATION.md §6:
    python tools/reference_pin/make_pin_ggufs.py /tmp/pin_ggufs          # here
    java ... GoldenDump /tmp/pin_ggufs/pin_llama_q8_0.gguf tests/golden/reference/pin_llama_q8_0.bin 4 4      # in the reference checkout
and commits tests/golden/reference/*.bin.  With a dump present these tests compare, bit for bit, the dump's logits of every step, the
greedy ids, the final x and the last position's K / V rows with (a) the C oracle in the mode the dump's header records and (b) the
HIP path (-m gpu).  Without dumps they SKIP with the reason "parity unpinned" — they never pass vacuously.
The dump format is written by tools/reference_pin/GoldenDump.java; test_dump_format_round_trip checks the parser on a synthetic
dump produced from the oracle itself (format only — that is not a pin)."""
import importlib.util
import os
import struct

import numpy as np
import pytest

import __graft_entry__ as ge

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, "golden", "reference")
_spec = importlib.util.spec_from_file_location("make_pin_ggufs", os.path.join(os.path.dirname(HERE), "tools", "reference_pin", "make_pin_ggufs.py"))
pin = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(pin)


def parse_dump(path):
    b = open(path, "rb").read()
    assert b[:8] == b"GL3REF01", "not a GoldenDump file"
    steps, vocab, dim, layers, kv_dim, n_prompt, vbits, qact = struct.unpack_from("<8i", b, 8)
    off = 8 + 32
    tokens = np.frombuffer(b, "<i4", steps + 1, off); off += 4 * (steps + 1)
    logits = np.frombuffer(b, "<f4", steps * vocab, off).reshape(steps, vocab); off += 4 * steps * vocab
    x = np.frombuffer(b, "<f4", dim, off); off += 4 * dim
    k = np.frombuffer(b, "<f4", layers * kv_dim, off).reshape(layers, kv_dim); off += 4 * layers * kv_dim
    v = np.frombuffer(b, "<f4", layers * kv_dim, off).reshape(layers, kv_dim); off += 4 * layers * kv_dim
    assert off == len(b), "trailing bytes"
    return dict(steps=steps, vocab=vocab, n_prompt=n_prompt, vector_bits=vbits, quantize_activation=bool(qact), tokens=tokens, logits=logits, x=x, k=k, v=v)


def write_dump(path, d):
    with open(path, "wb") as f:
        f.write(b"GL3REF01")
        f.write(struct.pack("<8i", d["steps"], d["vocab"], len(d["x"]), d["k"].shape[0], d["k"].shape[1], d["n_prompt"], d["vector_bits"], int(d["quantize_activation"])))
        for a, t in ((d["tokens"], "<i4"), (d["logits"], "<f4"), (d["x"], "<f4"), (d["k"], "<f4"), (d["v"], "<f4")):
            f.write(np.ascontiguousarray(a, t).tobytes())


def oracle_modes(d):
    """Oracle mode of a dump: the int8-activation Q8_0 dot is scalar whatever the species (Q8_0FloatTensor.java:90-123); the F16 /
    Q4_0 / f32-activation dots follow llama.VectorBitSize — r5: every species a JVM can report (0, 128, 256, 512) has a counterpart in
    both oracles (a dump of Q4_0 / Q8_0-f32act at 512 cannot exist: the reference throws there)."""
    assert d["vector_bits"] in (0, 128, 256, 512), d["vector_bits"]
    return dict(vector_bits=d["vector_bits"], f32_activation=not d["quantize_activation"])


def hip_flags(hip, modes, wt):
    """Plan flags of a dump's mode, or None when the library refuses the mode (128-bit species of a species-dependent type)."""
    species_type = wt in (1, 2) or modes["f32_activation"]
    vb = modes["vector_bits"]
    if vb == 128 and species_type:
        return None
    flags = hip.FLAG_F32_ACTIVATION if modes["f32_activation"] else 0
    if species_type:
        flags |= hip.FLAG_SCALAR_DOT if vb == 0 else hip.FLAG_VECTOR_512 if vb == 512 else 0
    return flags


def cases():
    out = []
    for stem, wt, seed, n_prompt, n_greedy, _flags in pin.PIN_CASES:
        out.append(pytest.param(stem, wt, seed, id=stem))
    return out


def load_case(pkg, stem, wt, seed):
    path = os.path.join(REF_DIR, stem + ".bin")
    if not os.path.exists(path):
        pytest.skip("parity unpinned: no reference dump %s (INTEGRATION.md §6: make_pin_ggufs.py + GoldenDump.java need a JDK)" % os.path.relpath(path, os.path.dirname(HERE)))
    d = parse_dump(path)
    m = pkg.synth.make_numpy(pin.pin_config(pkg), wtype=wt, seed=seed)
    assert d["vocab"] == m.cfg.vocab
    return d, m


def run_oracle(orc, m, d, **modes):
    o = orc.COracle(m, **modes)
    steps = d["steps"]
    toks = list(d["tokens"][:d["n_prompt"]])
    logits = []
    for pos in range(steps):
        lg = o.forward(int(toks[pos]), pos)
        logits.append(lg.copy())
        if pos >= d["n_prompt"] - 1:
            toks.append(orc.argmax(lg))
    return np.array(toks, np.int32), np.stack(logits), o


@pytest.mark.parametrize("stem,wt,seed", cases())
def test_c_oracle_matches_the_reference_dump(pkg, orc, stem, wt, seed):
    d, m = load_case(pkg, stem, wt, seed)
    toks, logits, o = run_oracle(orc, m, d, **oracle_modes(d))
    assert np.array_equal(toks, d["tokens"])
    assert np.array_equal(logits, d["logits"])
    for l in range(m.cfg.n_layers):
        k, v = o.kv(l, d["steps"] - 1)
        assert np.array_equal(k, d["k"][l]) and np.array_equal(v, d["v"][l])


@pytest.mark.gpu
@pytest.mark.parametrize("stem,wt,seed", cases())
def test_hip_path_matches_the_reference_dump(pkg, stem, wt, seed):
    from importlib import import_module
    d, m = load_case(pkg, stem, wt, seed)
    modes = oracle_modes(d)
    plan_mod, hip = import_module(ge.PKG_NAME + ".plan"), import_module(ge.PKG_NAME + ".hip")
    flags = hip_flags(hip, modes, wt)
    if flags is None:
        pytest.skip("dump made with the 128-bit species: restated in the oracles, refused by the library (GL3_FLAG_VECTOR_128)")
    plan = plan_mod.HipMasterPlan(m, flags=flags)
    for pos in range(d["steps"]):
        lg = plan.tornadoVMForwardDecode(int(d["tokens"][pos]), pos)
        assert np.array_equal(lg, d["logits"][pos]), pos
        if pos >= d["n_prompt"] - 1:
            assert int(np.argmax(lg)) == d["tokens"][pos + 1]
    for l in range(m.cfg.n_layers):
        k, v = plan.kv(l, d["steps"] - 1)
        assert np.array_equal(k, d["k"][l]) and np.array_equal(v, d["v"][l])
    plan.freeTornadoExecutionPlan()


def test_dump_format_round_trip(pkg, orc, tmp_path):
    """Format check only: a dump written from the ORACLE parses back and passes the comparison — this pins nothing."""
    base = pkg.synth.CONFIGS["tiny-llama"]
    m = pkg.synth.make_numpy(base, wtype=8, seed=7)
    d0 = dict(steps=5, vocab=m.cfg.vocab, n_prompt=3, vector_bits=256, quantize_activation=True,
              tokens=np.array(pkg.javarand.bench_tokens(m.cfg.vocab, 3) + [0, 0, 0], np.int32))
    toks, logits, o = run_oracle(orc, m, d0, vector_bits=0, f32_activation=False)
    k = np.stack([o.kv(l, 4)[0] for l in range(m.cfg.n_layers)])
    v = np.stack([o.kv(l, 4)[1] for l in range(m.cfg.n_layers)])
    d0.update(tokens=toks, logits=logits, x=np.zeros(m.cfg.dim, np.float32), k=k, v=v)
    p = str(tmp_path / "x.bin")
    write_dump(p, d0)
    d1 = parse_dump(p)
    assert np.array_equal(d1["tokens"], toks) and np.array_equal(d1["logits"], logits) and np.array_equal(d1["k"], k)
    assert d1["vector_bits"] == 256 and d1["quantize_activation"] and d1["n_prompt"] == 3


def test_pin_ggufs_are_loadable_and_carry_the_reference_loader_keys(pkg, tmp_path):
    """The files make_pin_ggufs.py writes parse with this repo's reader and hold what the reference's Llama loader reads: 'llama' in
    general.name, llama.* hyper-parameters, 128256 token strings with <|begin_of_text|> at 128000, an (empty) merges list."""
    base = pkg.synth.CONFIGS["tiny-llama"]
    cfg = pkg.synth.ModelConfig(**{**base.__dict__, "name": "pin-llama-random", "vocab": 128256, "ctx": 64, "n_layers": 1})
    m = pkg.synth.make_numpy(cfg, wtype=8, seed=7)
    md = m.metadata()
    md.update(pin.tokenizer_metadata(cfg.vocab))
    ts = [(n, [c] if r == 1 and ty == pkg.gguf.GGML_F32 else [c, r], ty, raw) for n, (raw, ty, r, c) in m.tensors.items()]
    p = str(tmp_path / "pin.gguf")
    pkg.gguf.write_gguf(p, md, ts)
    g = pkg.gguf.GGUFFile(p)
    assert "llama" in g.metadata["general.name"].lower() and g.metadata["llama.vocab_size"] == 128256
    toks = g.metadata["tokenizer.ggml.tokens"]
    assert len(toks) == 128256 and toks[128000] == "<|begin_of_text|>" and toks[128009] == "<|eot_id|>" and list(g.metadata["tokenizer.ggml.merges"]) == []
    assert g.tensors["token_embd.weight"][0] == [256, 128256]
