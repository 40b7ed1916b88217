"""Decode at depth for kernel traces: N layers of a named config, an untimed batched prefill of D positions, then T decode steps at
positions D .. (llama-bench -d).    python scripts/depth_only.py llama-3-8b 8 16384 16"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
from importlib import import_module
plan_mod = import_module(ge.PKG_NAME + ".plan")
name = sys.argv[1] if len(sys.argv) > 1 else "llama-3-8b"
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 8
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
ntok = int(sys.argv[4]) if len(sys.argv) > 4 else 16
base = pkg.synth.CONFIGS[name]
cfg = pkg.synth.ModelConfig(**{**base.__dict__, "n_layers": nl, "ctx": depth + ntok + 8})
m = pkg.synth.make_torch(cfg, seed=1, device="cuda")
plan = plan_mod.HipMasterPlan.initializeTornadoVMPlan(m, prefill_batch_size=512)
toks = pkg.javarand.bench_tokens(cfg.vocab, depth + ntok)
t0 = time.perf_counter()
plan.prefill(toks[:depth], 0)
torch.cuda.synchronize()
print("prefill of %d positions x %d layers: %.2f s" % (depth, nl, time.perf_counter() - t0), flush=True)
for i in range(2):
    plan.forward_decode(toks[depth + i], depth + i, copy=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(2, ntok):
    plan.forward_decode(toks[depth + i], depth + i, copy=False)
dt = (time.perf_counter() - t0) / (ntok - 2)
print("tg @ d%d, %d layers: %.1f us / token, %.2f us / layer (logits + embedding included)" % (depth, nl, dt * 1e6, dt * 1e6 / nl))
