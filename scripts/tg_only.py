"""tg timing only (full-size model, decode graph), for rocprofv3 kernel traces: python scripts/tg_only.py [model] [n_tokens]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
from importlib import import_module
plan_mod = import_module(ge.PKG_NAME + ".plan")
name = sys.argv[1] if len(sys.argv) > 1 else "llama-3-8b"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
base = pkg.synth.CONFIGS[name]
cfg = pkg.synth.ModelConfig(**{**base.__dict__, "ctx": 648})
m = pkg.synth.StreamModel(cfg, 8, pkg.synth.iter_torch(cfg, seed=1, device="cuda"))
plan = plan_mod.HipMasterPlan.initializeTornadoVMPlan(m, prefill_batch_size=1)
toks = pkg.javarand.bench_tokens(cfg.vocab, n)
for i in range(n):
    plan.forward_decode(toks[i], i, copy=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n):
    plan.forward_decode(toks[i], i, copy=False)
dt = time.perf_counter() - t0
print("tg%d %s: %.3f ms/token -> %.1f tok/s" % (n, name, dt / n * 1e3, n / dt))
