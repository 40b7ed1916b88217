"""world_size = 2 over gloo on CPU: the tensor-parallel scheme (row split of every matrix + all-gathers,
tests/tp_layout.py = the test-side description of the library's split rules) evaluated with the oracle's arithmetic is
bit-identical to the unsplit forward pass, and the host-side plumbing bench.py uses for N > 1 (id broadcast, streamed model) works across processes."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, cfg_name, out_q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    from importlib import import_module
    from oracle import oracle_np as onp
    pkg = ge.load_package()
    import tp_layout as tp
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = pkg.synth.CONFIGS[cfg_name]
    # every rank streams the same seeded model, as bench.py does for N > 1
    m = pkg.synth.StreamModel(cfg, pkg.synth.GGML_Q8_0, pkg.synth.iter_torch(cfg, seed=5))
    tensors = {k: v for k, v in m.tensor_items()}
    obj = [b"id-from-rank-0" if rank == 0 else None]
    dist.broadcast_object_list(obj, src=0)
    assert obj[0] == b"id-from-rank-0"
    sl = tp.row_slices(cfg, world, rank)

    def mm(name, x, d1):                     # this rank's rows of a Q8_0 matmul, in the oracle's arithmetic
        raw, ty, rows, cols = tensors[name]
        r0, n = sl[name.split(".", 2)[-1] if name.startswith("blk.") else name]
        rb = cols // 32 * 34
        return onp.matmul(raw[r0 * rb:(r0 + n) * rb], ty, x, n, d1)

    def gather(part):
        t = torch.from_numpy(np.ascontiguousarray(part))
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        return torch.cat(outs).numpy()

    hs, H, KVH = cfg.head_size, cfg.n_heads // world, cfg.n_kv_heads // world
    kvmul = cfg.n_heads // cfg.n_kv_heads
    cr, ci = m.rope
    kc = np.zeros((cfg.n_layers, cfg.ctx, KVH * hs), np.float32)
    vc = np.zeros_like(kc)
    toks = pkg.javarand.bench_tokens(cfg.vocab, 4)
    logits_all = []
    for pos, tok in enumerate(toks):
        raw, ty = tensors["token_embd.weight"][:2]
        x = onp.dequant(raw[tok * cfg.dim // 32 * 34:(tok + 1) * cfg.dim // 32 * 34], ty, cfg.dim)
        half = hs // 2
        fcr, fci = cr[pos * half:(pos + 1) * half], ci[pos * half:(pos + 1) * half]

        def rot(vec):
            vv = vec.reshape(-1, half, 2)
            o = np.empty_like(vv)
            o[:, :, 0] = vv[:, :, 0] * fcr - vv[:, :, 1] * fci
            o[:, :, 1] = vv[:, :, 0] * fci + vv[:, :, 1] * fcr
            return o.reshape(-1)
        for l in range(cfg.n_layers):
            p = f"blk.{l}."
            xb = onp.rmsnorm(x, onp.dequant(tensors[p + "attn_norm.weight"][0], 0, cfg.dim), cfg.rms_eps)
            q, k, v = rot(mm(p + "attn_q.weight", xb, cfg.dim)), rot(mm(p + "attn_k.weight", xb, cfg.dim)), mm(p + "attn_v.weight", xb, cfg.dim)
            kc[l, pos], vc[l, pos] = k, v
            att = np.zeros(H * hs, np.float32)
            for h in range(H):
                kk = kc[l, :pos + 1, (h // kvmul) * hs:(h // kvmul + 1) * hs]
                a = onp.softmax(onp.seq_sum(kk * q[h * hs:(h + 1) * hs][None, :], axis=1) / np.float32(np.sqrt(np.float64(hs))))
                att[h * hs:(h + 1) * hs] = onp.seq_sum(a[:, None] * vc[l, :pos + 1, (h // kvmul) * hs:(h // kvmul + 1) * hs], axis=0)
            xb_full = gather(att)                                                        # all-gather 1
            dl = cfg.dim // world
            x = x + mm(p + "attn_output.weight", xb_full, cfg.q_dim)                      # Wo is replicated: every rank, all rows, no gather
            xb = onp.rmsnorm(x, onp.dequant(tensors[p + "ffn_norm.weight"][0], 0, cfg.dim), cfg.rms_eps)
            g, u = mm(p + "ffn_gate.weight", xb, cfg.dim), mm(p + "ffn_up.weight", xb, cfg.dim)
            hb = gather(((g / (1.0 + np.exp(-g.astype(np.float64))).astype(np.float32)) * u).astype(np.float32))   # all-gather 2
            x = gather(x[rank * dl:(rank + 1) * dl] + mm(p + "ffn_down.weight", hb, cfg.hidden))   # all-gather 3
        xn = onp.rmsnorm(x, onp.dequant(tensors["output_norm.weight"][0], 0, cfg.dim), cfg.rms_eps)
        logits_all.append(gather(mm("output.weight", xn, cfg.dim)))
    if rank == 0:
        out_q.put(np.stack(logits_all))
    dist.barrier()
    dist.destroy_process_group()


def test_row_split_all_gather_scheme_is_bit_identical_world2(pkg):
    import torch.multiprocessing as mp
    from oracle import oracle_np as onp
    cfg_name = "tiny-llama"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, cfg_name, q)) for r in range(2)]
    [p.start() for p in procs]
    got = q.get(timeout=240)
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    m = pkg.synth.make_torch(pkg.synth.CONFIGS[cfg_name], seed=5)
    o = onp.NpOracle(m.oracle_cfg(), m.oracle_tensors(), m.rope)
    for pos, t in enumerate(pkg.javarand.bench_tokens(m.cfg.vocab, 4)):
        assert np.array_equal(got[pos], o.forward(t, pos)), pos


def test_partition_table_matches_library_rules(pkg):
    from importlib import import_module
    import __graft_entry__ as ge
    import tp_layout as tp
    c = pkg.synth.CONFIGS["llama-3-8b"]
    for n in (1, 2, 4, 8):
        s = [tp.row_slices(c, n, r) for r in range(n)]
        assert all(x["attn_output.weight"] == (0, c.dim) for x in s)              # replicated
        for name, full in (("attn_q.weight", c.q_dim), ("attn_k.weight", c.kv_dim),
                           ("ffn_gate.weight", c.hidden), ("ffn_down.weight", c.dim), ("output.weight", c.vocab)):
            assert sum(x[name][1] for x in s) == full and all(x[name][0] % 16 == 0 for x in s)
            assert [x[name][0] for x in s] == [r * full // n for r in range(n)]
    with pytest.raises(ValueError):
        tp.validate(c, 3)
    with pytest.raises(ValueError):
        tp.validate(c, 16)            # only 8 kv heads


def _chunk_worker(rank, world, port, out_q):
    """The in-place all-gather of a rank-chunked prefill activation, over gloo: each rank fills only its chunk of the flat
    [tp][ntok][cols/tp] buffer with its rows of a row-split product; after all_gather_into_tensor every rank must hold the
    full [ntok][cols] activation when read through tp.chunked_index."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    from importlib import import_module
    ge.load_package()
    import tp_layout as tp
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ntok, cols, k = 5, 96, 64
    cc = cols // world
    rng = np.random.default_rng(7)
    w = rng.standard_normal((cols, k)).astype(np.float32)          # same on every rank
    x = rng.standard_normal((ntok, k)).astype(np.float32)
    mine = x @ w[rank * cc:(rank + 1) * cc].T                      # [ntok][cc]: this rank's rows of the product
    buf = torch.zeros(world * ntok * cc)
    buf[rank * ntok * cc:(rank + 1) * ntok * cc] = torch.from_numpy(np.ascontiguousarray(mine).reshape(-1))
    dist.all_gather_into_tensor(buf, buf[rank * ntok * cc:(rank + 1) * ntok * cc].clone())
    full = tp.from_chunked(buf.numpy(), world, ntok, cols)
    ok = bool(np.array_equal(full, np.concatenate([x @ w[r * cc:(r + 1) * cc].T for r in range(world)], axis=1)))
    probe = all(buf[tp.chunked_index(b, j, cc, ntok)].item() == full[b, j] for b in range(ntok) for j in (0, cc - 1, cc, cols - 1))
    if rank == 0:
        out_q.put((bool(np.array_equal(full[:, rank * cc:(rank + 1) * cc], mine)), probe, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_rank_chunked_prefill_gather_world2(pkg):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_chunk_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    own, probe, ok = q.get(timeout=120)
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert own and probe and ok


def test_chunked_layout_round_trip(pkg):
    from importlib import import_module
    import __graft_entry__ as ge
    import tp_layout as tp
    x = np.arange(7 * 24, dtype=np.float32).reshape(7, 24)
    for n in (1, 2, 4):
        flat = tp.to_chunked(x, n)
        assert np.array_equal(tp.from_chunked(flat, n, 7, 24), x)
        assert all(flat[tp.chunked_index(b, j, 24 // n, 7)] == x[b, j] for b in range(7) for j in range(24))
    c = pkg.synth.CONFIGS["llama-3-8b"]
    assert tp.prefill_gather_points(c, 8, 512) == [("AO", 512 * 512), ("HB", 512 * 1792), ("X", 512 * 512)]
