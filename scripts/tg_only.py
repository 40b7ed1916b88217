"""Decode-only driver for kernel traces: N layers of a named config, T greedy tokens from position 0 (graph replay).
    python scripts/tg_only.py llama-3-8b 8 2 64 [f32act]      # model, layers, ggml type (8 Q8_0 / 1 F16 / 2 Q4_0), tokens"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
from importlib import import_module
plan_mod = import_module(ge.PKG_NAME + ".plan")
name = sys.argv[1] if len(sys.argv) > 1 else "llama-3-8b"
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 8
wtype = int(sys.argv[3]) if len(sys.argv) > 3 else 8
ntok = int(sys.argv[4]) if len(sys.argv) > 4 else 64
flags = pkg.hip.FLAG_F32_ACTIVATION if len(sys.argv) > 5 and sys.argv[5] == "f32act" else 0
base = pkg.synth.CONFIGS[name]
cfg = pkg.synth.ModelConfig(**{**base.__dict__, "n_layers": nl, "ctx": max(256, ntok + 8)})
m = pkg.synth.make_torch(cfg, wtype=wtype, seed=1, device="cuda")
plan = plan_mod.HipMasterPlan.initializeTornadoVMPlan(m, flags=flags)
tok = 1
for p in range(4):
    tok = plan.forward_decode_argmax(tok, p)
torch.cuda.synchronize()
t0 = time.perf_counter()
for p in range(4, 4 + ntok):
    tok = plan.forward_decode_argmax(tok, p)
dt = (time.perf_counter() - t0) / ntok
print("tg %d layers: %.1f us / token, %.2f us / layer (logits + embedding included)" % (nl, dt * 1e6, dt * 1e6 / nl))
