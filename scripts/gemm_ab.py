"""Batched-prefill GEMM classes at 512 tokens on N layers of a named config: one HIP event pair per class (gl3_profile_prefill_kernel)
+ the pp512 wall time.  A/B switches are environment variables read by the library (GL3_PF_GEMM2, GL3_PF_GEMM2_OCC, ...).
    python scripts/gemm_ab.py llama-3-8b 4"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
from importlib import import_module
plan_mod = import_module(ge.PKG_NAME + ".plan")
name = sys.argv[1] if len(sys.argv) > 1 else "llama-3-8b"
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ntok = int(sys.argv[3]) if len(sys.argv) > 3 else 512
base = pkg.synth.CONFIGS[name]
cfg = pkg.synth.ModelConfig(**{**base.__dict__, "n_layers": nl, "vocab": 4096, "ctx": 648})
m = pkg.synth.make_torch(cfg, wtype=8, seed=1, device="cuda")
plan = plan_mod.HipMasterPlan.initializeTornadoVMPlan(m, prefill_batch_size=512)
toks = pkg.javarand.bench_tokens(cfg.vocab, ntok)
plan.prefill(toks, 0)
out = []
for k in ("matvec_qkv", "matvec_wo", "matvec_gateup", "matvec_down"):
    r = plan.profile_prefill_kernel(k, ntok, iters=5)
    out.append("%s %.1f us (%.0f TOP/s)" % (k.replace("matvec_", ""), r["avg_us"], r["tops"]))
t0 = time.perf_counter()
for _ in range(3):
    plan.prefill(toks, 0)
dt = (time.perf_counter() - t0) / 3
tag = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("GL3_"))
print("[%s] %s | pp%d %d layers: %.2f ms = %.1f us/layer" % (tag, "; ".join(out), ntok, nl, dt * 1e3, dt * 1e6 / nl))
