#!/usr/bin/env python3
"""Writes the small GGUF files that tools/reference_pin/GoldenDump.java feeds to the REAL reference (pure-Java CPU path), so that a
maintainer with a JDK can pin this repo's oracle:   python tools/reference_pin/make_pin_ggufs.py /tmp/pin_ggufs

The models are tests/test_reference_golden.py's PIN_CASES: the tiny-llama shape (dim 256, 2 layers, 8 / 2 heads of 32) under the REAL
Llama-3 vocabulary size, because the reference's loader needs it: LlamaTokenizer treats ids >= 128000 as special tokens and
Llama.createNewState looks up "<|begin_of_text|>" (tokenizer/LlamaTokenizer.java:47-51, model/llama/Llama.java:51), and
ModelLoader.detectModelType wants "llama" in general.name (model/loader/ModelLoader.java:65).  Weights are this repo's bit-stable
NumPy Philox stream (synth.make_numpy), so the test regenerates them from (config, type, seed) instead of reading the file."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

SPECIAL = {0: "<|begin_of_text|>", 1: "<|end_of_text|>", 6: "<|start_header_id|>", 7: "<|end_header_id|>", 9: "<|eot_id|>"}


def pin_config(pkg):
    base = pkg.synth.CONFIGS["tiny-llama"]
    return pkg.synth.ModelConfig(**{**base.__dict__, "name": "pin-llama-random", "vocab": 128256, "ctx": 64})


# (file stem, ggml type, seed, prompt tokens, greedy steps, JVM flags the dump must be produced with)
PIN_CASES = [
    ("pin_llama_q8_0", 8, 7, 4, 4, ""),                                   # default: int8 activation, scalar everywhere
    ("pin_llama_f16_scalar", 1, 7, 4, 4, "-Dllama.VectorBitSize=0"),
    ("pin_llama_f16_v256", 1, 7, 4, 4, "-Dllama.VectorBitSize=256"),
    ("pin_llama_q8_0_f32act_v256", 8, 7, 4, 4, "-Dllama.VectorBitSize=256 -Dllama.quantizeActivation=false"),
    # r5: the other species a JVM can report (FloatTensor.java:21) — 512 is what an AVX-512 host such as the GPU box's EPYC 9575F picks by
    # default; F16 is species-generic, the Q8_0 vector dot has a 128-bit branch (and throws at 512: no dump can exist)
    ("pin_llama_f16_v512", 1, 7, 4, 4, "-Dllama.VectorBitSize=512"),
    ("pin_llama_f16_v128", 1, 7, 4, 4, "-Dllama.VectorBitSize=128"),
    ("pin_llama_q8_0_f32act_v128", 8, 7, 4, 4, "-Dllama.VectorBitSize=128 -Dllama.quantizeActivation=false"),
]


def tokenizer_metadata(vocab):
    toks = ["t%d" % i for i in range(128000)]
    toks += [SPECIAL.get(i, "<|reserved_special_token_%d|>" % i) for i in range(vocab - 128000)]
    return {"tokenizer.ggml.model": "gpt2", "tokenizer.ggml.tokens": toks, "tokenizer.ggml.merges": []}


def main(out_dir):
    pkg = ge.load_package()
    os.makedirs(out_dir, exist_ok=True)
    cfg = pin_config(pkg)
    done = {}
    for stem, wt, seed, n_prompt, n_greedy, flags in PIN_CASES:
        gguf_name = "pin_llama_%s.gguf" % {8: "q8_0", 1: "f16"}[wt]
        if gguf_name not in done:
            m = pkg.synth.make_numpy(cfg, wtype=wt, seed=seed)
            md = m.metadata()
            md.update(tokenizer_metadata(cfg.vocab))
            ts = []
            for name, (raw, ty, rows, cols) in m.tensors.items():
                ts.append((name, [cols] if rows == 1 and ty == pkg.gguf.GGML_F32 else [cols, rows], ty, raw))
            pkg.gguf.write_gguf(os.path.join(out_dir, gguf_name), md, ts)
            done[gguf_name] = True
        print("java --enable-preview --add-modules jdk.incubator.vector %s -cp <pin classes>:<reference classes>:<tornado api jars> GoldenDump %s "
              "tests/golden/reference/%s.bin %d %d" % (flags, os.path.join(out_dir, gguf_name), stem, n_prompt, n_greedy))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/tmp/pin_ggufs")
