"""Debug helper: tensor-parallel batched prefill of a VALU-GEMM type (Q4_0 / Q8_0 with the f32 activation) with the ranks as
threads of ONE process, repeated, reporting where the KV cache first differs from the oracle and what the wrong rows hold.
    DBG_ITERS=10 DBG_ORDER=q4,f32 [GL3_TP_DEBUG=1] python scripts/debug_tp_vl.py
GL3_TP_DEBUG=1 switches the gather to its checksummed twin (gl3_tp.hip): a transport fault prints "[gl3 tp dbg]" lines."""
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

_cache = {}


def oracle_kv(cfg, chunks, wtype, f32act):
    key = (cfg, tuple(chunks), wtype, f32act)
    if key not in _cache:
        m = pkg.synth.make_numpy(pkg.synth.CONFIGS[cfg], wtype=wtype, seed=41)
        o = orc.COracle(m, vector_bits=0 if (wtype == 8 and not f32act) else 256, f32_activation=f32act)
        n = sum(chunks)
        toks = pkg.javarand.bench_tokens(m.cfg.vocab, n + 2)
        o.prefill(toks[:n], 0)
        kv = [[o.kv(l, p) for p in range(n)] for l in range(m.cfg.n_layers)]
        _cache[key] = (m, toks, kv)
    return _cache[key]


def run(cfg, tp, chunks, wtype, f32act, tag):
    m, toks, okv = oracle_kv(cfg, chunks, wtype, f32act)
    n = sum(chunks)
    kvl = m.cfg.kv_dim // tp
    grp = plan_mod.make_local_group(tp)
    kvs, kvs2, err = [None] * tp, [None] * tp, [None] * tp
    bar = threading.Barrier(tp)
    sync_first = os.environ.get("DBG_BARRIER", "0") == "1"      # read the KV cache only after EVERY rank has finished computing

    def rank_main(r):
        try:
            plan = plan_mod.HipMasterPlan(m, prefill_batch_size=64, tp_rank=r, tp_size=tp, local_group=grp, flags=hip.FLAG_F32_ACTIVATION if f32act else 0)
            pos = 0
            for c in chunks:
                plan.tornadoVMForwardBatchPrefill(toks[pos:pos + c], pos); pos += c
            if sync_first: bar.wait()
            kvs[r] = [[plan.kv(l, p) for p in range(n)] for l in range(m.cfg.n_layers)]
            bar.wait()                                   # second read with the device idle: tells a wrong READBACK from a wrong cache
            kvs2[r] = [[plan.kv(l, p) for p in range(n)] for l in range(m.cfg.n_layers)]
            bar.wait()
            plan.freeTornadoExecutionPlan()
        except Exception as e:
            err[r] = e
            bar.abort()
    th = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(tp)]
    [t.start() for t in th]; [t.join(timeout=300) for t in th]
    hip.lib().gl3_local_group_destroy(grp)
    if any(err):
        print(tag, "ERR", err, flush=True); return -1
    bad = []
    for r in range(tp):
        for l in range(m.cfg.n_layers):
            for p in range(n):
                ko, vo = okv[l][p]
                k, v = kvs[r][l][p]
                if not np.array_equal(k, ko[r * kvl:(r + 1) * kvl]) or not np.array_equal(v, vo[r * kvl:(r + 1) * kvl]):
                    bad.append((r, l, p))
    reread = sum(1 for (r, l, p) in bad if not (np.array_equal(kvs[r][l][p][0], kvs2[r][l][p][0]) and np.array_equal(kvs[r][l][p][1], kvs2[r][l][p][1])))
    print(tag, "mismatching (rank, layer, pos):", len(bad), bad[:12], "| rows whose second read (device idle) differs from the first:", reread, flush=True)
    if bad:
        # the earliest layer with a wrong row tells where the fault entered; is the wrong row some OTHER (layer, pos) row of the oracle?
        l0 = min(b[1] for b in bad)
        for (r, l, p) in [b for b in bad if b[1] == l0][:4]:
            k, v = kvs[r][l][p]
            ko, vo = okv[l][p]
            nd = int(np.sum(k != ko[r * kvl:(r + 1) * kvl]))
            dk = np.nonzero(k != ko[r * kvl:(r + 1) * kvl])[0]
            print("      differing k indices %d..%d, k row all zero: %s, v row equal: %s" % (dk.min(), dk.max(), not k.any(), np.array_equal(v, vo[r * kvl:(r + 1) * kvl])), flush=True)
            twin = [(l2, p2) for l2 in range(m.cfg.n_layers) for p2 in range(n) if np.array_equal(k, okv[l2][p2][0][r * kvl:(r + 1) * kvl])]
            print("   first bad layer %d: rank %d pos %d: %d of %d k elements differ, max |diff| %.3g; equals oracle k row of (layer, pos) %s" %
                  (l, r, p, nd, kvl, float(np.max(np.abs(k - ko[r * kvl:(r + 1) * kvl]))), twin[:3]), flush=True)
    return len(bad)


order = os.environ.get("DBG_ORDER", "q4,f32").split(",")
iters = int(os.environ.get("DBG_ITERS", "2"))
fails = {}
t0 = time.time()
for it in range(iters):
    for o_ in order:
        if o_ == "int8": nb = run("mid-llama", 2, [40, 9], 8, False, "int8 tp2 it%d" % it)
        elif o_ == "q4": nb = run("mid-llama", 4, [33, 20], 2, False, "q4_0 tp4 it%d" % it)
        elif o_ == "q4b": nb = run("mid-llama", 2, [33, 20], 2, False, "q4_0 tp2 it%d" % it)
        elif o_ == "f32": nb = run("mid-llama", 2, [50, 9], 8, True, "q8 f32act tp2 it%d" % it)
        elif o_ == "f16": nb = run("mid-llama", 2, [40, 9], 1, False, "f16 tp2 it%d" % it)
        else: continue
        f = fails.setdefault(o_, [0, 0]); f[1] += 1; f[0] += 1 if nb else 0
print("SUMMARY (failed / runs):", {k: "%d/%d" % tuple(v) for k, v in fails.items()}, "in %.0f s" % (time.time() - t0), flush=True)
