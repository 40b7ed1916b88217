import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
from importlib import import_module
plan_mod = import_module(ge.PKG_NAME + ".plan"); hip = import_module(ge.PKG_NAME + ".hip")
from oracle import oracle_np
def rel(a, b): return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))
base = pkg.synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "mid-llama"]
cfg = pkg.synth.ModelConfig(**{**base.__dict__, "n_layers": 1})
m = pkg.synth.make_numpy(cfg, seed=21)
plan = plan_mod.HipMasterPlan(m, flags=hip.FLAG_LAYER_TAPS)
o = oracle_np.NpOracle(m.oracle_cfg(), m.oracle_tensors(), m.rope)
toks = pkg.javarand.bench_tokens(cfg.vocab, 6)
cap = {}
orig = o._mm
def mm(name, x, d0, d1):
    r = orig(name, x, d0, d1)
    key = name.split(".")[-2]
    cap[key + "_in"] = x.copy(); cap[key + "_out"] = r.copy()
    return r
o._mm = mm
for pos, t in enumerate(toks):
    ref = o.forward(t, pos)
    got = plan.tornadoVMForwardDecode(t, pos)
    qkv = plan.buffer(0, cfg.q_dim + 2 * cfg.kv_dim)
    xb = plan.buffer(1, cfg.q_dim); hb = plan.buffer(2, cfg.hidden)
    q, k, v = qkv[:cfg.q_dim], qkv[cfg.q_dim:cfg.q_dim + cfg.kv_dim], qkv[cfg.q_dim + cfg.kv_dim:]
    print("pos", pos, "q %.1e k %.1e v %.1e" % (rel(q, cap["attn_q_out"]), rel(k, cap["attn_k_out"]), rel(v, cap["attn_v_out"])),
          "xb %.1e" % rel(xb, cap["attn_output_in"]), "hb %.1e" % rel(hb, cap["ffn_down_in"]), "logits %.1e" % rel(got, ref),
          "kc %.1e vc %.1e" % (rel(plan.kv(0, pos)[0], o.kc[0, pos]), rel(plan.kv(0, pos)[1], o.vc[0, pos])), flush=True)
    # recompute attention on the host from the GPU's own cache to isolate the attention kernel
    H, hs, kvm = cfg.n_heads, cfg.head_size, cfg.n_heads // cfg.n_kv_heads
    K = np.stack([plan.kv(0, p)[0] for p in range(pos + 1)]); V = np.stack([plan.kv(0, p)[1] for p in range(pos + 1)])
    # roped q from oracle formula applied to GPU raw q
    half = hs // 2; cr = m.rope[0][pos * half:(pos + 1) * half]; ci = m.rope[1][pos * half:(pos + 1) * half]
    qq = q.reshape(-1, half, 2); qr = np.empty_like(qq); qr[:, :, 0] = qq[:, :, 0] * cr - qq[:, :, 1] * ci; qr[:, :, 1] = qq[:, :, 0] * ci + qq[:, :, 1] * cr
    qr = qr.reshape(H, hs)
    exp_xb = np.zeros(cfg.q_dim, np.float32)
    for h in range(H):
        kk = K[:, (h // kvm) * hs:(h // kvm + 1) * hs].astype(np.float64); vv = V[:, (h // kvm) * hs:(h // kvm + 1) * hs].astype(np.float64)
        s = kk @ qr[h].astype(np.float64) / np.sqrt(hs); a = np.exp(s - s.max()); a /= a.sum()
        exp_xb[h * hs:(h + 1) * hs] = a @ vv
    print("      attention kernel vs f64 recompute from GPU q/cache: %.1e" % rel(xb, exp_xb), flush=True)
