"""tg128 of a Qwen1.5-MoE-A2.7B-shaped random-weight model (GL3_ARCH_QWEN2MOE, Q8_0) on one MI355X.

The reference's LlamaBench protocol for tg (128 decode steps from an empty cache, logits to the host every step), the per-class
device times of instrumented steps, and the algorithmic bytes one token streams: attention matrices, router, the 4 selected
experts, the shared expert, norms, KV cache, the vocabulary projection.  Prints one JSON line.

    python scripts/moe_tg.py [--layers 24] [--steps 3]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=24)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--n-gen", type=int, default=128)
    args = ap.parse_args()
    import torch
    import __graft_entry__ as ge
    from importlib import import_module
    pkg = ge.load_package()
    synth = pkg.synth
    plan_mod, hip = import_module(ge.PKG_NAME + ".plan"), import_module(ge.PKG_NAME + ".hip")
    cfg = synth.CONFIGS["qwen1.5-moe-a2.7b"]
    cfg = synth.ModelConfig(**{**cfg.__dict__, "ctx": args.n_gen + 8, "n_layers": args.layers})
    t0 = time.time()
    mdl = synth.StreamModel(cfg, synth.GGML_Q8_0, synth.iter_torch(cfg, wtype=synth.GGML_Q8_0, seed=42, device="cuda"))
    plan = plan_mod.HipMasterPlan(mdl)
    setup_s = time.time() - t0
    toks = pkg.javarand.bench_tokens(cfg.vocab, args.n_gen)

    def rep():
        for i in range(args.n_gen):
            plan.forward_decode(toks[i], i, copy=False)
    rep()
    torch.cuda.synchronize()
    samples = []
    for _ in range(args.steps):
        t1 = time.perf_counter()
        rep()
        samples.append(time.perf_counter() - t1)
    tok_s = args.steps * args.n_gen / sum(samples)
    acc = None
    n_prof = 8
    for i in range(n_prof):
        k = plan.profile_decode(toks[64 + i], 64 + i)
        if acc is None:
            acc = k
        else:
            for name in k:
                for f in ("ms", "launches", "bytes"):
                    acc[name][f] += k[name][f]
    classes = {n: dict(us_per_token=round(v["ms"] / n_prof * 1e3, 1), launches_per_token=v["launches"] // n_prof,
                       gbs=round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["ms"] > 0 else None) for n, v in acc.items() if v["launches"]}
    q8 = 34 / 32
    L, d, E, k, mh, sh = cfg.n_layers, cfg.dim, cfg.n_experts, cfg.n_experts_used, cfg.moe_hidden, cfg.hidden
    per_layer = (cfg.q_dim * d + 2 * cfg.kv_dim * d + d * cfg.q_dim) * q8 + 3 * k * mh * d * q8 + 3 * sh * d * q8 + (E + 1) * d * 4 + 2 * d * 4
    avg_pos = (args.n_gen - 1) / 2.0
    token_bytes = int(L * per_layer + cfg.vocab * d * q8 + d * q8 + d * 4 + 2 * L * cfg.kv_dim * 4 * (avg_pos + 2) + cfg.vocab * 4)
    resident = int(L * ((cfg.q_dim * d * 2 + 2 * cfg.kv_dim * d) * q8 + 3 * E * mh * d * q8 + 3 * sh * d * q8) + 2 * cfg.vocab * d * q8)
    print(json.dumps(dict(metric="tg%d tokens/s" % args.n_gen, value=round(tok_s, 2), unit="tokens/s", n_gpus=1, steps=args.steps, dtype="i8",
                          data="synthetic", config=dict(workload="Qwen1.5-MoE-A2.7B shape, Q8_0, %d layers, 60 experts top-4, random weights" % L),
                          ms_per_token=round(1e3 / tok_s, 3), samples_tok_s=[round(args.n_gen / s, 2) for s in samples],
                          algorithmic_bytes_per_token=token_bytes, effective_gbs=round(token_bytes * tok_s / 1e9, 1),
                          frac_of_hbm_peak=round(token_bytes * tok_s / 8e12, 4), resident_weight_bytes=resident, kernel_classes=classes,
                          setup_s=round(setup_s, 1))))


if __name__ == "__main__":
    main()
