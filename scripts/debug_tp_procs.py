"""Debug helper: the production tensor-parallel layout (one PROCESS per rank, IPC-mapped arenas, all ranks on device 0 of the
one-GPU box) looped N times inside the same processes: plan -> batched prefill (16, 16, 5) -> 7 decode steps -> free, every
iteration compared with the CPU oracle.
    python scripts/debug_tp_procs.py <world> <ggml type> <iterations> [f32act]"""
import os, sys, socket, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def worker(rank, world, port, wtype, f32act, iters, ref, q):
    import torch  # noqa: F401
    import torch.distributed as dist
    import __graft_entry__ as ge
    from importlib import import_module
    pkg = ge.load_package()
    plan_mod = import_module(ge.PKG_NAME + ".plan"); hip = import_module(ge.PKG_NAME + ".hip")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def exchange(handle):
        out = [None] * world
        dist.all_gather_object(out, handle)
        return out
    m = pkg.synth.make_numpy(pkg.synth.CONFIGS["mid-llama"], wtype=wtype, seed=17)
    toks = pkg.javarand.bench_tokens(m.cfg.vocab, 44)
    bad = 0
    for it in range(iters):
        plan = plan_mod.HipMasterPlan(m, prefill_batch_size=16, tp_rank=rank, tp_size=world, p2p_exchange=exchange,
                                      flags=hip.FLAG_F32_ACTIVATION if f32act else 0)
        dist.barrier()
        plan.prefill(toks[:37], 0)
        out = [plan.forward_decode(toks[p], p) for p in range(37, 44)]
        dist.barrier()
        plan.freeTornadoExecutionPlan()
        wrong = [i for i in range(7) if not np.array_equal(out[i], ref[i])]
        if wrong:
            bad += 1
            print("rank %d iteration %d: decode steps %s differ from the oracle" % (rank, it, wrong), flush=True)
        dist.barrier()
    q.put((rank, bad))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    wtype = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    f32act = len(sys.argv) > 4 and sys.argv[4] == "f32act"
    import __graft_entry__ as ge
    pkg = ge.load_package()
    from oracle import oracle_c as orc
    orc.build()
    m = pkg.synth.make_numpy(pkg.synth.CONFIGS["mid-llama"], wtype=wtype, seed=17)
    o = orc.COracle(m, vector_bits=0 if (wtype == 8 and not f32act) else 256, f32_activation=f32act)
    toks = pkg.javarand.bench_tokens(m.cfg.vocab, 44)
    o.prefill(toks[:37], 0)
    ref = [o.forward(toks[p], p).copy() for p in range(37, 44)]
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    t0 = time.time()
    procs = [ctx.Process(target=worker, args=(r, world, port, wtype, f32act, iters, ref, q)) for r in range(world)]
    [p.start() for p in procs]
    res = dict(q.get(timeout=1500) for _ in range(world))
    [p.join(timeout=120) for p in procs]
    print("SUMMARY procs world %d type %d%s: iterations with a wrong result per rank %s of %d, exit codes %s, %.0f s" %
          (world, wtype, " f32act" if f32act else "", res, iters, [p.exitcode for p in procs], time.time() - t0), flush=True)
