import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
from importlib import import_module
plan_mod = import_module(ge.PKG_NAME + ".plan"); hip = import_module(ge.PKG_NAME + ".hip")
from oracle import oracle_c
def rel(a, b): return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))
for cfg in sys.argv[1:] or ["mid-llama"]:
    m = pkg.synth.make_numpy(pkg.synth.CONFIGS[cfg], seed=21)
    plan = plan_mod.HipMasterPlan(m, flags=hip.FLAG_LAYER_TAPS)
    o = oracle_c.COracle(m)
    toks = pkg.javarand.bench_tokens(m.cfg.vocab, 6)
    for pos, t in enumerate(toks):
        ref, lx = o.forward(t, pos, layer_x=True)
        got = plan.tornadoVMForwardDecode(t, pos)
        errs = [rel(plan.layer_x(l), lx[l]) for l in range(m.cfg.n_layers)]
        kv = []
        for l in range(m.cfg.n_layers):
            k, v = plan.kv(l, pos); ko, vo = o.kv(l, pos)
            kv.append((rel(k, ko), rel(v, vo)))
        print(cfg, "pos", pos, "logits %.2e" % rel(got, ref), "layer_x", ["%.1e" % e for e in errs], "kv", [("%.1e" % a, "%.1e" % b) for a, b in kv], flush=True)
