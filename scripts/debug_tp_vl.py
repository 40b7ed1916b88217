"""Debug helper: tensor-parallel batched prefill of a VLQ type in-process, repeated, reporting where the KV cache first differs from the oracle."""
import sys, os, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
from importlib import import_module
plan_mod = import_module(ge.PKG_NAME + ".plan"); hip = import_module(ge.PKG_NAME + ".hip")
from oracle import oracle_c as orc
orc.build()

def run(cfg, tp, chunks, wtype, f32act, tag):
    m = pkg.synth.make_numpy(pkg.synth.CONFIGS[cfg], wtype=wtype, seed=41)
    o = orc.COracle(m, vector_bits=0 if (wtype == 8 and not f32act) else 256, f32_activation=f32act)
    n = sum(chunks)
    toks = pkg.javarand.bench_tokens(m.cfg.vocab, n + 2)
    o.prefill(toks[:n], 0)
    kvl = m.cfg.kv_dim // tp
    grp = plan_mod.make_local_group(tp)
    kvs, err = [None] * tp, [None] * tp
    def rank_main(r):
        try:
            plan = plan_mod.HipMasterPlan(m, prefill_batch_size=64, tp_rank=r, tp_size=tp, local_group=grp, flags=hip.FLAG_F32_ACTIVATION if f32act else 0)
            pos = 0
            for c in chunks:
                plan.tornadoVMForwardBatchPrefill(toks[pos:pos + c], pos); pos += c
            kvs[r] = [[plan.kv(l, p) for p in range(n)] for l in range(m.cfg.n_layers)]
            plan.freeTornadoExecutionPlan()
        except Exception as e:
            err[r] = e
    th = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(tp)]
    [t.start() for t in th]; [t.join(timeout=300) for t in th]
    hip.lib().gl3_local_group_destroy(grp)
    if any(err): print(tag, "ERR", err); return
    bad = []
    for r in range(tp):
        for l in range(m.cfg.n_layers):
            for p in range(n):
                ko, vo = o.kv(l, p)
                k, v = kvs[r][l][p]
                if not np.array_equal(k, ko[r * kvl:(r + 1) * kvl]) or not np.array_equal(v, vo[r * kvl:(r + 1) * kvl]):
                    bad.append((r, l, p))
    print(tag, "mismatching (rank, layer, pos):", len(bad), bad[:12])

order = os.environ.get("DBG_ORDER", "q4,f32").split(",")
for it in range(2):
    for o_ in order:
        if o_ == "int8": run("mid-llama", 2, [40, 9], 8, False, "int8 tp2")
        if o_ == "q4": run("mid-llama", 4, [33, 20], 2, False, "q4_0 tp4 it%d" % it)
        if o_ == "q4b": run("mid-llama", 2, [33, 20], 2, False, "q4_0 tp2 it%d" % it)
        if o_ == "f32": run("mid-llama", 2, [50, 9], 8, True, "q8 f32act tp2 it%d" % it)
        if o_ == "f16": run("mid-llama", 2, [40, 9], 1, False, "f16 tp2 it%d" % it)
