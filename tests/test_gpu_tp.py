"""Tensor parallelism on ONE GPU: the in-process test transport (gl3_local_group: one host thread per rank,
device-to-device copies instead of RCCL) runs the real row-split plans — same upload slicing, same kernel arguments,
same gather points — and every rank must return logits bit-identical to the single-GPU CPU oracle.
A one-rank RCCL communicator additionally exercises ncclCommInitRank / ncclAllGather inside the library."""
import threading

import numpy as np
import pytest

import __graft_entry__ as ge

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def planmod():
    from importlib import import_module
    ge.load_package()
    return import_module(ge.PKG_NAME + ".plan"), import_module(ge.PKG_NAME + ".hip")


@pytest.mark.parametrize("cfg,tp,wtype", [("mid-llama", 2, 8), ("mid-llama", 4, 8), ("mid-qwen3", 2, 8), ("tiny-llama-tied", 2, 8),
                                          ("mid-llama", 2, 2), ("mid-llama", 4, 1), ("mid-qwen3", 2, 1)])   # BASELINE configs[3]: Q4_0 row split
def test_row_split_ranks_are_bit_identical_to_the_oracle(pkg, orc, planmod, cfg, tp, wtype):
    plan_mod, hip = planmod
    m = pkg.synth.make_numpy(pkg.synth.CONFIGS[cfg], wtype=wtype, seed=17)
    o = orc.COracle(m)
    toks = pkg.javarand.bench_tokens(m.cfg.vocab, 6)
    ref = [o.forward(t, p) for p, t in enumerate(toks)]
    grp = plan_mod.make_local_group(tp)
    out = [None] * tp
    err = [None] * tp

    def rank_main(r):
        try:
            plan = plan_mod.HipMasterPlan(m, tp_rank=r, tp_size=tp, local_group=grp)
            out[r] = [plan.forward_decode(t, p) for p, t in enumerate(toks)]
            plan.freeTornadoExecutionPlan()
        except Exception as e:   # noqa: BLE001
            err[r] = e

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(tp)]
    [t.start() for t in th]
    [t.join(timeout=120) for t in th]
    assert all(e is None for e in err), err
    assert not any(t.is_alive() for t in th)
    hip.lib().gl3_local_group_destroy(grp)
    for r in range(tp):
        for p in range(len(toks)):
            assert np.array_equal(out[r][p], ref[p]), (r, p)


def test_single_rank_rccl_communicator(pkg, orc, planmod):
    plan_mod, hip = planmod
    m = pkg.synth.make_numpy(pkg.synth.CONFIGS["tiny-llama"], seed=7)
    o = orc.COracle(m)
    uid = plan_mod.make_unique_id()
    plan = plan_mod.HipMasterPlan(m, flags=hip.FLAG_FORCE_RCCL, unique_id=uid)      # tp_size = 1, all-gathers still issued
    for pos, t in enumerate(pkg.javarand.bench_tokens(m.cfg.vocab, 4)):
        assert np.array_equal(plan.forward_decode(t, pos), o.forward(t, pos))
    plan.freeTornadoExecutionPlan()
