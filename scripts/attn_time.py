import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
from importlib import import_module
plan_mod = import_module(ge.PKG_NAME + ".plan"); hip = import_module(ge.PKG_NAME + ".hip")
base = pkg.synth.CONFIGS["llama-3-8b"]
cfg = pkg.synth.ModelConfig(**{**base.__dict__, "n_layers": 4, "vocab": 4096})
m = pkg.synth.make_torch(cfg, seed=1, device="cuda")
plan = plan_mod.HipMasterPlan(m)
toks = pkg.javarand.bench_tokens(cfg.vocab, 640)
for pos in range(640):
    plan.forward_decode(toks[pos], pos, copy=False)
    if pos in (0, 31, 63, 127, 255, 639):
        import ctypes as C
        res = {}
        for name, k in (("scores", 5), ("softmax_pv", 6)):
            us, nb = C.c_double(), C.c_uint64()
            hip.check(hip.lib().gl3_profile_kernel(plan._ctx, k, 50, C.byref(us), C.byref(nb)), plan._ctx)
            res[name] = round(us.value, 2)
        print("pos", pos, res, flush=True)
