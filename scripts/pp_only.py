import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
from importlib import import_module
plan_mod = import_module(ge.PKG_NAME + ".plan")
name = sys.argv[1] if len(sys.argv) > 1 else "llama-3-8b"
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 8
wtype = int(sys.argv[3]) if len(sys.argv) > 3 else 8          # ggml type of the matrices: 8 = Q8_0, 1 = F16, 2 = Q4_0
depth = int(os.environ.get("PP_DEPTH", "0"))                  # llama-bench -d: the timed chunk sits behind `depth` untimed positions
base = pkg.synth.CONFIGS[name]
cfg = pkg.synth.ModelConfig(**{**base.__dict__, "n_layers": nl, "vocab": 4096, "ctx": 648 + depth})
m = pkg.synth.make_torch(cfg, wtype=wtype, seed=1, device="cuda")
plan = plan_mod.HipMasterPlan.initializeTornadoVMPlan(m, prefill_batch_size=512)
toks = pkg.javarand.bench_tokens(cfg.vocab, 512)
for off in range(0, depth, 512):
    plan.prefill(toks[:min(512, depth - off)], off)
plan.prefill(toks, depth)
t0 = time.perf_counter()
for _ in range(3):
    plan.prefill(toks, depth)
dt = (time.perf_counter() - t0) / 3
print("pp512%s %d layers: %.2f ms -> %.0f tok/s (x%d layers = %.1f ms / 32 layers)" % ("@d%d" % depth if depth else "", nl, dt * 1e3, 512 / dt, nl, dt * 1e3 * 32 / nl))
