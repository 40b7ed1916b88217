"""Step-by-step run of a full-size F16 / Q4_0 plan (debugging aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
from importlib import import_module
plan_mod = import_module(ge.PKG_NAME + ".plan"); hip = import_module(ge.PKG_NAME + ".hip"); synth = pkg.synth
name, wt = sys.argv[1], int(sys.argv[2])
cfg = synth.CONFIGS[name]
cfg = synth.ModelConfig(**{**cfg.__dict__, "ctx": 648})
dev = torch.device("cuda", 0)
model = synth.StreamModel(cfg, wt, synth.iter_torch(cfg, wtype=wt, seed=42, device=dev))
print("creating plan", flush=True)
plan = plan_mod.HipMasterPlan(model, prefill_batch_size=int(sys.argv[3]) if len(sys.argv) > 3 else 1, flags=hip.FLAG_NO_GRAPH)
print("plan ok", flush=True)
lg = plan.forward_decode(5, 0)
print("decode ok", float(np.abs(lg).max()), flush=True)
for k in ("matvec_qkv", "matvec_wo", "matvec_gateup", "matvec_down", "matvec_logits"):
    r = plan.profile_kernel(k, iters=3)
    print(k, r, flush=True)
print(plan.profile_decode(7, 1))
