"""Static batched decode timing only: python scripts/bd_only.py [model] [B] [steps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
from importlib import import_module
plan_mod = import_module(ge.PKG_NAME + ".plan")
name = sys.argv[1] if len(sys.argv) > 1 else "qwen3-4b"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
n = int(sys.argv[3]) if len(sys.argv) > 3 else 32
base = pkg.synth.CONFIGS[name]
cfg = pkg.synth.ModelConfig(**{**base.__dict__, "ctx": 136})
m = pkg.synth.StreamModel(cfg, 8, pkg.synth.iter_torch(cfg, seed=1, device="cuda"))
plan = plan_mod.HipMasterPlan.initializeTornadoVMPlan(m, prefill_batch_size=B, n_seqs=B)
toks = np.asarray(pkg.javarand.bench_tokens(cfg.vocab, n * B), np.int32).reshape(n, B)
seqs = np.arange(B, dtype=np.int32)
for i in range(n):
    plan.forward_decode_batch(toks[i], seqs, np.full(B, i, np.int32), want_logits=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n):
    plan.forward_decode_batch(toks[i], seqs, np.full(B, i, np.int32), want_logits=False)
dt = time.perf_counter() - t0
print("batched decode %s B=%d: %.3f ms/step -> %.1f tok/s" % (name, B, dt / n * 1e3, B * n / dt))
