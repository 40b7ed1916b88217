"""Generates tests/golden/*.npz with the NumPy oracle (oracle/oracle_np.py).

PARITY UNPINNED: the reference has no golden vectors for the forward pass and
cannot run here, so these fixtures are produced by this repo's own restatement; they guard the
oracle against drift and give the HIP path fixed expected outputs.  Weights are NOT stored: they
are regenerated from the NumPy Philox seed (bit-stable), the fixture holds token ids, logits of
every step, x after every layer of the last step, and the greedy continuation.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
from oracle import oracle_np  # noqa: E402

CASES = [  # (fixture, config, ggml type, seed, prompt tokens, greedy steps, vector bits of the F16 / Q4_0 dots)
    ("tiny_llama_q8_0", "tiny-llama", 8, 7, 6, 24, 0),
    ("tiny_llama_f16", "tiny-llama", 1, 7, 4, 8, 0),                 # scalar dots (-Dllama.VectorBitSize=0)
    ("tiny_llama_tied_q4_0", "tiny-llama-tied", 2, 11, 4, 8, 0),
    ("tiny_qwen3_q8_0", "tiny-qwen3", 8, 5, 6, 24, 0),
    ("tiny_qwen2_q8_0", "tiny-qwen2", 8, 13, 5, 12, 0),
    ("tiny_granite_q8_0", "tiny-granite", 8, 19, 5, 12, 0),          # forwardGranite: llama graph + embedding / attention / residual / logit scalars
    ("tiny_phi3_q8_0", "tiny-phi3", 8, 23, 5, 12, 0),                # forwardJavaPhi3: fused attn_qkv / gate|up, NeoX RoPE
    ("tiny_devstral_q8_0", "tiny-devstral", 8, 29, 5, 12, 0),        # forwardJavaDevstral: head_size != dim / heads, YaRN table
    ("tiny_qwen2moe_q8_0", "tiny-qwen2moe", 8, 31, 5, 12, 0),        # forwardJavaQwen2MoE: router / top-2 of 8 experts / gated shared expert
    ("tiny_llama_f16_v256", "tiny-llama", 1, 7, 4, 8, 256),          # Vector-API dots, 256-bit species (the reference's default order)
    ("tiny_llama_tied_q4_0_v256", "tiny-llama-tied", 2, 11, 4, 8, 256),
    # Q8_0 weights with vector bits 256 = -Dllama.quantizeActivation=false: Q8_0FloatTensor.vectorDot on the f32 activation
    ("tiny_llama_q8_0_f32act_v256", "tiny-llama", 8, 7, 4, 8, 256),
]


def run_case(pkg, cfg_name, wtype, seed, n_prompt, n_greedy, vbits=0):
    m = pkg.synth.make_numpy(pkg.synth.CONFIGS[cfg_name], wtype=wtype, seed=seed)
    o = oracle_np.NpOracle(m.oracle_cfg(), m.oracle_tensors(), m.rope, vector_bits=vbits, f32_activation=(wtype == 8 and vbits == 256))
    prompt = pkg.javarand.bench_tokens(m.cfg.vocab, n_prompt)
    logits, lx = [], []
    tok_stream = list(prompt)
    for pos in range(n_prompt + n_greedy - 1):
        lx = []
        lg = o.forward(tok_stream[pos], pos, layer_x=lx)
        logits.append(lg)
        if pos >= n_prompt - 1:
            tok_stream.append(oracle_np.argmax(lg))
    return dict(prompt=np.array(prompt, np.int32), tokens=np.array(tok_stream, np.int32),
                logits=np.stack(logits).astype(np.float32), last_layer_x=np.stack(lx).astype(np.float32),
                k_last=o.kc[:, n_prompt + n_greedy - 2].copy(), v_last=o.vc[:, n_prompt + n_greedy - 2].copy())


if __name__ == "__main__":
    pkg = ge.load_package()
    only = sys.argv[1:]
    for fx, cfg_name, wt, seed, npmt, ng, vbits in CASES:
        if only and fx not in only:
            continue
        out = run_case(pkg, cfg_name, wt, seed, npmt, ng, vbits)
        np.savez_compressed(os.path.join(os.path.dirname(__file__), fx + ".npz"), **out)
        print(fx, out["tokens"].tolist())
