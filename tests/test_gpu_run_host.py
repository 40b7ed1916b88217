"""tools/gl3_run (plain C++ over the C-ABI) against the reference's generation protocols restated on the CPU oracle.

The (token, position) sequences are those of InferenceEngine.generateTokensGPULlama (J/inference/InferenceEngine.java:293-382: the
state's begin-of-text token first, then every prompt token, then the sampled ones), of its batched-prefill twin
(InferenceEngineWithBatchPrefillDecode.java:163-250) and of generateTokensGPUQwen3 (:383-475: no begin-of-text, and the position that
the loop skips after the last prompt token).  The Python loops below are an independent restatement over oracle.forward; the ids
the native host prints must be identical — greedy and sampled (same L32X64MixRandom coins)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


EXE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "gl3_run")


def llama_loop(orc, o, prompt, bos, max_tokens, stop, pick):
    gen, cur, pos, pi = [], bos, 0, 0
    while pos < max_tokens:
        logits = o.forward(cur, pos)
        if pi < len(prompt):
            nxt = prompt[pi]; pi += 1
        else:
            nxt = pick(logits)
            gen.append(nxt)
            if nxt in stop:
                break
        cur = nxt
        pos += 1
    return gen


def qwen3_loop(orc, o, prompt, max_tokens, stop, pick):
    gen, cur, pi, position = [], 0, 0, 0
    while position < max_tokens:
        if pi < len(prompt):
            logits = o.forward(prompt[pi], position)
            pi += 1
            if pi < len(prompt):
                position += 1
                continue
            position += 1                       # "the current logit belongs to the next position" (:431) ...
        else:
            logits = o.forward(cur, position)
        nxt = pick(logits)
        gen.append(nxt)
        if nxt in stop:
            break
        cur = nxt
        position += 1                           # ... and the loop's own ++position: one KV row is never written
    return gen


def run_host(path, prompt, extra):
    out = subprocess.run([EXE, "-m", path, "--ids", ",".join(map(str, prompt))] + extra, capture_output=True, text=True, timeout=180)
    assert out.returncode == 0, out.stderr
    return [int(x) for x in out.stdout.split("generated:")[1].split()], out.stderr


@pytest.mark.parametrize("cfg,batch", [("tiny-llama", 1), ("tiny-llama", 4), ("tiny-qwen3", 1)])
def test_greedy_generation_matches_the_reference_protocol(pkg, orc, tmp_path, cfg, batch):
    m = pkg.synth.make_numpy(pkg.synth.CONFIGS[cfg], seed=31)
    path = str(tmp_path / "m.gguf")
    m.write_gguf(path)
    prompt = pkg.javarand.bench_tokens(m.cfg.vocab, 7)
    o = orc.COracle(pkg.synth.SynthModel.from_gguf(path))
    if "qwen3" in cfg:
        want = qwen3_loop(orc, o, prompt, 24, set(), orc.argmax)
        got, err = run_host(path, prompt, ["-n", "24", "-b", str(batch)])
        assert "protocol qwen3" in err
    else:
        want = llama_loop(orc, o, prompt, 1, 24, set(), orc.argmax)
        got, err = run_host(path, prompt, ["-n", "24", "-b", str(batch), "--bos", "1"])
        assert "protocol llama" in err
    assert got == want and len(got) > 8


def test_stop_token_and_sampled_generation(pkg, orc, tmp_path):
    m = pkg.synth.make_numpy(pkg.synth.CONFIGS["tiny-llama"], seed=37)
    path = str(tmp_path / "m.gguf")
    m.write_gguf(path)
    prompt = pkg.javarand.bench_tokens(m.cfg.vocab, 5)
    o = orc.COracle(pkg.synth.SynthModel.from_gguf(path))
    greedy = llama_loop(orc, o, prompt, 1, 30, set(), orc.argmax)
    stop = greedy[4]                                                   # the fifth generated id stops the run and is part of the result
    o = orc.COracle(pkg.synth.SynthModel.from_gguf(path))
    want = llama_loop(orc, o, prompt, 1, 30, {stop}, orc.argmax)
    got, _ = run_host(path, prompt, ["-n", "30", "--bos", "1", "--stop", "%d,%d" % (stop, 999999)])
    assert got == want and got[-1] == stop and len(got) <= 5
    # temperature / top-p: one rng.nextFloat(1f) per sampled token, L32X64MixRandom(seed) as RandomGeneratorFactory.getDefault()
    for temperature, topp in ((0.8, 0.9), (1.1, 0.0)):
        rng = pkg.javarand.L32X64MixRandom(4321)
        o = orc.COracle(pkg.synth.SynthModel.from_gguf(path))
        want = llama_loop(orc, o, prompt, 1, 22, set(), lambda lg: orc.sample(lg, temperature, topp, rng.next_float()))
        got, _ = run_host(path, prompt, ["-n", "22", "--bos", "1", "--temperature", str(temperature), "--top-p", str(topp), "--seed", "4321"])
        assert got == want, (temperature, topp)
