"""Model check of the tensor-parallel hand-over protocols of the decode step (gpullama3.java_amd/csrc/gl3_tp.hip, "folded gathers"; flag / step
arithmetic of TpRec in gl3_decode_kernels.h and tp_fold_setup in gl3_api.hip), CPU only.

Every rank runs the launches of its decode steps in stream order; ranks interleave arbitrarily (random schedules).  A producer launch writes its
own chunk of a gathered buffer and — at some later point of the schedule, but before it publishes — the peers' copies; consumers read.  Every
read asserts that each chunk holds exactly the version the reference's data flow needs (too old = a missing wait, too new = a peer's later push
landed before this rank's last read of the previous version: the write-after-read hazard the protocols claim to exclude WITHOUT acknowledgements).
A schedule in which no rank can advance is a dead-lock.  Modes: 0 = one gather kernel per hand-over (push + flag + wait, global sequence number),
1 = producers push, a wait launch precedes the consumer, 2 = the wait is the consumer's own prologue (the embedding kernel waits for the previous
step's last pushes into x).  Steps with and without the logits projection (prefill tokens skip it, so nothing joins the ranks at a step's end)."""
import random

import pytest

XB, X, HB, LG = 0, 1, 2, 3


class Rank:
    def __init__(self, r, tp, L):
        self.r, self.tp, self.L = r, tp, L
        self.buf = {b: [None] * tp for b in (XB, X, HB, LG)}      # version held by this rank's copy of chunk c of buffer b
        self.flag = {b: [0] * tp for b in (XB, X, HB)}            # folded protocol: gathers of buffer b completed by rank p into this arena
        self.gflag = [0] * tp                                     # gather-kernel protocol: gathers completed by rank p (all buffers)
        self.seq = 0                                              # gather kernels run by this rank
        self.step = 0
        self.pc = 0
        self.prog = []


def build_program(rk, mode, steps):
    """List of (kind, payload) launches in stream order.  kind: 'op' (reads, local writes), 'push' (remote writes + flag), 'wait'."""
    L, tp, me = rk.L, rk.tp, rk.r
    prog = []

    def gather_kernel(buf, version):       # modes 0 (all buffers) and every mode for the logits: push own chunk + flag, then wait for the peers
        prog.append(("gpush", (buf, version)))
        prog.append(("gwait", None))

    for s, want_logits in enumerate(steps, start=1):
        def k_of(add): return (s - 1) * L + add
        # embedding: (mode 2) waits for the previous step's last pushes into x, then overwrites x
        prog.append(("op", dict(name="embed", wait=[(X, k_of(0))] if mode == 2 else [], reads=[], writes=[(X, c, ("E", s)) for c in range(tp)], bump_step=True)))
        for l in range(L):
            x_in = ("E", s) if l == 0 else ("D", s, l - 1)
            prog.append(("op", dict(name="qkv", wait=[(X, k_of(l))] if (mode == 2 and l > 0) else [], reads=[(X, c, x_in) for c in range(tp)], writes=[])))
            prog.append(("op", dict(name="attn", wait=[], reads=[], writes=[(XB, me, ("A", s, l))])))
            if mode == 0: gather_kernel(XB, ("A", s, l))
            else:
                prog.append(("push", (XB, ("A", s, l), k_of(l + 1))))
                if mode == 1: prog.append(("wait", (XB, k_of(l + 1))))
            prog.append(("op", dict(name="wo", wait=[(XB, k_of(l + 1))] if mode == 2 else [],
                                    reads=[(XB, c, ("A", s, l)) for c in range(tp)] + [(X, c, x_in) for c in range(tp)],
                                    writes=[(X, c, ("W", s, l)) for c in range(tp)])))                       # Wo is replicated: every rank rewrites all of x
            prog.append(("op", dict(name="gateup", wait=[], reads=[(X, c, ("W", s, l)) for c in range(tp)], writes=[(HB, me, ("H", s, l))])))
            if mode == 0: gather_kernel(HB, ("H", s, l))
            else:
                prog.append(("push", (HB, ("H", s, l), k_of(l + 1))))
                if mode == 1: prog.append(("wait", (HB, k_of(l + 1))))
            prog.append(("op", dict(name="down", wait=[(HB, k_of(l + 1))] if mode == 2 else [],
                                    reads=[(HB, c, ("H", s, l)) for c in range(tp)] + [(X, me, ("W", s, l))], writes=[(X, me, ("D", s, l))])))
            if mode == 0: gather_kernel(X, ("D", s, l))
            else:
                prog.append(("push", (X, ("D", s, l), k_of(l + 1))))
                if mode == 1: prog.append(("wait", (X, k_of(l + 1))))
        if want_logits:
            prog.append(("op", dict(name="logits", wait=[(X, k_of(L))] if mode == 2 else [], reads=[(X, c, ("D", s, L - 1)) for c in range(tp)],
                                    writes=[(LG, me, ("L", s))])))
            gather_kernel(LG, ("L", s))
            prog.append(("op", dict(name="sample", wait=[], reads=[(LG, c, ("L", s)) for c in range(tp)], writes=[])))
    return prog


def runnable(rk, ranks):
    kind, p = rk.prog[rk.pc]
    if kind == "wait":
        buf, k = p
        return all(rk.flag[buf][q] >= k for q in range(rk.tp) if q != rk.r)
    if kind == "gwait":
        return all(rk.gflag[q] >= rk.seq for q in range(rk.tp) if q != rk.r)
    if kind == "op":
        return all(rk.flag[buf][q] >= k for (buf, k) in p["wait"] for q in range(rk.tp) if q != rk.r)
    return True


def execute(rk, ranks):
    kind, p = rk.prog[rk.pc]
    if kind == "op":
        for (buf, c, want) in p["reads"]:
            got = rk.buf[buf][c]
            assert got == want, "rank %d %s: chunk %d of buffer %d holds %r, needs %r" % (rk.r, p["name"], c, buf, got, want)
        for (buf, c, v) in p["writes"]:
            rk.buf[buf][c] = v
        if p.get("bump_step"): rk.step += 1
    elif kind == "push":                    # folded producer: the peers' copies, then the per-buffer flag
        buf, v, k = p
        assert (rk.step - 1) * rk.L < k <= rk.step * rk.L                      # k is derived from the device step word the embedding kernel bumped
        for q in ranks:
            if q.r != rk.r:
                q.buf[buf][rk.r] = v
                q.flag[buf][rk.r] = k
    elif kind == "gpush":                   # gather kernel: the peers' copies, then the global sequence number
        buf, v = p
        rk.seq += 1
        for q in ranks:
            if q.r != rk.r:
                q.buf[buf][rk.r] = v
                q.gflag[rk.r] = rk.seq
    rk.pc += 1


def run(mode, tp, L, steps, seed):
    rng = random.Random(seed)
    ranks = [Rank(r, tp, L) for r in range(tp)]
    for rk in ranks: rk.prog = build_program(rk, mode, steps)
    # biased schedules: some ranks run far ahead whenever they can (the write-after-read hazard needs a fast producer and a slow reader)
    bias = [rng.choice((1, 1, 5, 25)) for _ in range(tp)]
    while any(rk.pc < len(rk.prog) for rk in ranks):
        ready = [rk for rk in ranks if rk.pc < len(rk.prog) and runnable(rk, ranks)]
        assert ready, "dead-lock: " + str([(rk.r, rk.prog[rk.pc][0]) for rk in ranks if rk.pc < len(rk.prog)])
        rk = rng.choices(ready, weights=[bias[q.r] for q in ready])[0]
        execute(rk, ranks)


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("tp,L", [(2, 1), (2, 3), (4, 2), (8, 2)])
def test_no_stale_or_early_chunk_and_no_deadlock(mode, tp, L):
    for seed in range(150):
        steps = [bool((seed >> i) & 1) or i == 3 for i in range(4)]          # mixes prefill steps (no logits) and sampled steps
        run(mode, tp, L, steps, seed)


def test_the_model_sees_a_missing_wait():
    """Sanity of the checker itself: mode 2 without the embedding kernel's wait lets a peer's last push of step s land on step s + 1's embedding."""
    def broken(rk, mode, steps):
        prog = build_program(rk, mode, steps)
        return [(k, dict(p, wait=[]) if (k == "op" and p["name"] == "embed") else p) for (k, p) in prog]
    hit = 0
    for seed in range(300):
        rng = random.Random(seed)
        ranks = [Rank(r, 2, 1) for r in range(2)]
        for rk in ranks: rk.prog = broken(rk, 2, [False, False, False])
        try:
            while any(rk.pc < len(rk.prog) for rk in ranks):
                ready = [rk for rk in ranks if rk.pc < len(rk.prog) and runnable(rk, ranks)]
                assert ready
                execute(rng.choices(ready, weights=[1 if q.r else 25 for q in ready])[0], ranks)
        except AssertionError:
            hit += 1
    assert hit > 0
