"""Tensor-parallel decode with folded gathers, ranks as threads of ONE fresh process, every rank's logits compared with the CPU oracle bit for bit
(used by tests/test_gpu_tp.py: a fresh process has one hardware queue per rank stream, see tests/conftest.py).
    GL3_TP_FOLD=2 [GL3_TP_FOLD_MASK=m] python scripts/debug_tp_fold.py mid-llama 4 [tokens] [ggml type: 8 Q8_0 | 2 Q4_0 | 1 F16]"""
import sys, os, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
from importlib import import_module
plan_mod = import_module(ge.PKG_NAME + ".plan"); hip = import_module(ge.PKG_NAME + ".hip")
from oracle import oracle_c as orc
orc.build()
cfg, tp = sys.argv[1], int(sys.argv[2])
ntok = int(sys.argv[3]) if len(sys.argv) > 3 else 3
wtype = int(sys.argv[4]) if len(sys.argv) > 4 else 8
m = pkg.synth.make_numpy(pkg.synth.CONFIGS[cfg], wtype=wtype, seed=17)
o = orc.COracle(m, vector_bits=0 if wtype == 8 else 256)
toks = pkg.javarand.bench_tokens(m.cfg.vocab, ntok)
ref = [o.forward(t, p) for p, t in enumerate(toks)]
grp = plan_mod.make_local_group(tp)
out, err = [None] * tp, [None] * tp
ready = threading.Barrier(tp)        # a host-level barrier between plan creation and the first forward

def rank_main(r):
    try:
        plan = plan_mod.HipMasterPlan(m, tp_rank=r, tp_size=tp, local_group=grp)
        if r == 0: print("fold mode %d mask %d" % plan.tp_fold_mode(), flush=True)
        ready.wait()
        res = []
        for p, t in enumerate(toks):
            t0 = time.time(); res.append(plan.forward_decode(t, p)); print("rank %d token %d: %.3f s" % (r, p, time.time() - t0), flush=True)
        out[r] = res
        plan.freeTornadoExecutionPlan()
    except Exception as e:
        err[r] = e
        ready.abort()

th = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(tp)]
t0 = time.time()
[t.start() for t in th]; [t.join(timeout=300) for t in th]
print("errors:", err, "in %.1f s" % (time.time() - t0))
bad = sum(e is not None for e in err)
for r in range(tp):
    if out[r] is None: bad += 1; continue
    eq = [bool(np.array_equal(out[r][p], ref[p])) for p in range(len(toks))]
    bad += eq.count(False)
    print("rank", r, eq)
sys.exit(1 if bad else 0)
