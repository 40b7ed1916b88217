#!/usr/bin/env python3
"""llama-bench-style pp512 / tg128 for the HIP forward pass (LlamaBench protocol, J/bench/LlamaBench.java:172-273).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): random-weight
Llama-3-8B-shaped Q8_0 model (dim 4096, hidden 14336, 32 layers, 32/8 heads, vocab 128256), token ids from
java.util.Random(42) as LlamaBench does, context = 512 + 128 + 8.

A *step* is one llama-bench repetition of tg128: 128 single-token decode steps at positions 0..127 over a
growing KV cache, each returning the full logits to host memory (the reference's timed region includes the
logits D2H, LlamaBench.java:234-254).  `value` = tokens / wall over exactly K steps (weights already in HBM).
pp512 (batched prefill, -b 512, no logits) is timed the same way and reported beside it.

pp512 is also reported at -b 128 and at -b 1 (LlamaBench: with -b 1 every prompt token is a full decode step
with logits + D2H).

N > 1 runs ONE model tensor-parallel over N GPUs, one process per GPU: every matrix is split by OUTPUT rows (heads /
hidden units / dim rows / vocab rows) so each dot product stays whole and in the reference's order on one rank
(bit-identical results), Wo is replicated, and the activations are re-assembled by 3 all-gathers per layer + 1 for the logits
(DESIGN.md section 7).  Total work is fixed, so "scaling": "strong".  Without a launcher `--gpus N` starts its own N ranks and
fails loudly if fewer than N devices are visible.

Extra objects: `roofline` — the dominant decode kernel (fused gate/up Q8_0 matvec): `avg_us` = one HIP event pair on the
plan's stream around back-to-back launches over all layers' weights (the figure the committed rocprofv3 kernel statistics
of this command agree with); `avg_us_instrumented_steps` = the kernel's own begin / end timestamps (hipExtLaunchKernel
start / stop events) in eager decode steps, reported beside it; `roofline_pp` —
the dominant batched-prefill kernel (gate/up int8-MFMA GEMM at 512 tokens); `cpu_baseline` — the C oracle
(oracle/gl3_oracle.c, kind "port") on the host cores, bounded sample, rank 0 at N=1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X spec sheet (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s measured achievable)
INT8_MFMA_PEAK_TOPS = 5000.0 # dense int8 MFMA (same rate class as FP8, ~5 P(FL)OP/s spec; micro-benchmark ceiling 3944 TOP/s)


def probe_peaks(device):
    """Measured-achievable peaks of this device (gl3_probe_peaks: 1 GiB streaming read / copy, int8 MFMA loop), SURVEY.md 8d."""
    import ctypes as C
    from importlib import import_module
    import __graft_entry__ as ge
    ge.load_package()
    hip = import_module(ge.PKG_NAME + ".hip")
    rd, cp, tops = C.c_double(), C.c_double(), C.c_double()
    rc = hip.lib().gl3_probe_peaks(device, C.byref(rd), C.byref(cp), C.byref(tops))
    if rc != 0:
        return None
    return dict(hbm_read_gbs=round(rd.value, 1), hbm_copy_gbs=round(cp.value, 1), int8_mfma_tops=round(tops.value, 1),
                method="gl3_probe_peaks in this run: best of 3 — streaming float4 read of 1 GiB, device-to-device copy of 1 GiB (read + written bytes), "
                       "4 independent v_mfma_i32_32x32x32_i8 accumulators per wavefront on every SIMD")


def host_cpu_info():
    """CPU model and the affinity mask the CPU baseline ran under."""
    model = None
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        aff = sorted(os.sched_getaffinity(0))
        runs, start, prev = [], aff[0], aff[0]
        for c in aff[1:] + [None]:
            if c is None or c != prev + 1:
                runs.append("%d-%d" % (start, prev) if prev != start else "%d" % start)
                start = c
            prev = c
        mask = ",".join(runs)
    except (AttributeError, OSError, IndexError):
        aff, mask = [], None
    return dict(cpu_model=model, affinity=mask, affinity_cpus=len(aff), logical_cpus=os.cpu_count())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)        # llama-bench -r 5
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="llama-3-8b")
    ap.add_argument("--n-gen", type=int, default=128)
    ap.add_argument("--n-prompt", type=int, default=512)
    ap.add_argument("--batch", type=int, default=512, help="prefill chunk (llama-bench -b)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pp", action="store_true")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--wtype", default="q8_0", choices=["q8_0", "f16", "q4_0", "q8_0_f32act"],
                    help="ggml type of the matrices (default: the headline Q8_0); q8_0_f32act = Q8_0 with -Dllama.quantizeActivation=false "
                         "(f32 activation, Q8_0FloatTensor.vectorDot)")
    ap.add_argument("--depth", default="", help="comma list of context depths (llama-bench -d): after the headline measurement, tg<n-gen> is timed again "
                    "behind an untimed batched prefill of d positions (LlamaBench.runTest :233-254) and reported as depth_rows")
    ap.add_argument("--decode-batch", type=int, default=0, help="BASELINE configs[4]: static-batched decode of B independent sequences "
                    "(e.g. --model qwen3-4b --decode-batch 32); prints its own JSON line instead of the tg/pp line")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_spawn(args.gpus)

    import numpy as np
    import torch
    import __graft_entry__ as ge
    pkg = ge.load_package()
    from importlib import import_module
    plan_mod = import_module(ge.PKG_NAME + ".plan")
    hip = import_module(ge.PKG_NAME + ".hip")
    synth = pkg.synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch one rank per GPU, or run without WORLD_SIZE and let "
                         "bench.py start the ranks itself)" % (args.gpus, world))
    if world > 1 and not os.environ.get("GL3_BENCH_SHARE_GPU") and torch.cuda.device_count() < world:
        raise SystemExit("bench.py: --gpus %d but only %d device(s) are visible to rank %d" % (world, torch.cuda.device_count(), rank))
    dist = None
    transport = os.environ.get("GL3_TP_TRANSPORT", "p2p")        # p2p: peer-write all-gather over xGMI (csrc/gl3_tp.hip); rccl: fall-back
    if os.environ.get("GL3_BENCH_SHARE_GPU"):                    # test hook: every rank on device 0 of a one-GPU box (IPC between processes)
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        # control plane only (handle / id exchange, barriers, max-over-ranks of the timing): gloo.  The data path between the
        # GPUs is the library's own transport; RCCL is initialised inside the library when GL3_TP_TRANSPORT=rccl.
        dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    depths = [int(x) for x in args.depth.split(",") if x.strip()]
    cfg = synth.CONFIGS[args.model]
    cfg = synth.ModelConfig(**{**cfg.__dict__, "ctx": max([args.n_prompt + args.n_gen] + [d + args.n_gen for d in depths]) + 8})   # LlamaBench: max(depth+tokens)+8
    if args.decode_batch > 0:
        return bench_decode_batch(args, cfg, synth, plan_mod, pkg, torch, np, dev)
    t0 = time.time()
    keep_host = (world == 1 and not args.no_cpu_baseline)
    wtype = {"q8_0": synth.GGML_Q8_0, "f16": synth.GGML_F16, "q4_0": synth.GGML_Q4_0, "q8_0_f32act": synth.GGML_Q8_0}[args.wtype]
    WT = args.wtype.upper()
    bpe = {"q8_0": 34 / 32, "f16": 2.0, "q4_0": 18 / 32, "q8_0_f32act": 34 / 32}[args.wtype]          # weight bytes per element
    plan_flags = hip.FLAG_F32_ACTIVATION if args.wtype == "q8_0_f32act" else 0
    toks = pkg.javarand.bench_tokens(cfg.vocab, max([args.n_prompt + args.n_gen] + [d + args.n_gen for d in depths]))

    def build_plan(transport):
        uid, exchange = None, None
        if world > 1 and transport == "rccl":
            obj = [plan_mod.make_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(obj, src=0)
            uid = obj[0]
        elif world > 1:
            def exchange(handle):
                out = [None] * world
                dist.all_gather_object(out, handle)
                return out
        if keep_host:
            mdl = synth.make_torch(cfg, wtype=wtype, seed=args.seed, device=dev)
        else:
            mdl = synth.StreamModel(cfg, wtype, synth.iter_torch(cfg, wtype=wtype, seed=args.seed, device=dev))
        pl = plan_mod.HipMasterPlan.initializeTornadoVMPlan(mdl, prefill_batch_size=args.batch, device=local_rank,
                                                             tp_rank=rank, tp_size=world, unique_id=uid, p2p_exchange=exchange, flags=plan_flags)
        if dist is not None:
            dist.barrier()                  # every rank's plan exists and is attached before the first gather
        return mdl, pl

    transport = select_transport(transport, world, rank, local_rank, dist, pkg, plan_mod, hip, torch)
    model, plan = build_plan(transport)
    torch.cuda.empty_cache()
    setup_s = time.time() - t0

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def tg_rep():
        for i in range(args.n_gen):
            plan.forward_decode(toks[i], i, copy=False)

    def pp_rep(batch=None):
        if batch == 1:                      # LlamaBench -b 1: every prompt token is a full decode step (logits + D2H)
            for i in range(args.n_prompt):
                plan.forward_decode(toks[i], i, copy=False)
        else:
            plan.prefill(toks[:args.n_prompt], 0, batch=batch)

    def timed(fn, steps, warmup):
        import gc
        gc.collect()                            # the timed window measures the forward calls, not a generation-2 collection of the host model's objects
        gc.disable()                            # (seen as one repetition of five 12 % slower whenever the host copy of the weights was alive)
        for _ in range(warmup):                 # warm-up directly in front of the timed window: the collection above idles the device for a few hundred
            fn()                                # milliseconds, and the first 15 ms prefill repetition behind such a pause ran 10 % below the others
        samples = []
        barrier()
        t_start = time.perf_counter()
        for _ in range(steps):
            t1 = time.perf_counter()
            fn()
            samples.append(time.perf_counter() - t1)
        barrier()
        total = time.perf_counter() - t_start
        gc.enable()
        if dist is not None:
            t = torch.tensor([total], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            total = float(t.item())
        return total, samples

    tg_total, tg_samples = timed(tg_rep, args.steps, args.warmup)
    tg_tok_s = args.steps * args.n_gen / tg_total
    pp, pp_rows = None, []
    if not args.no_pp:
        for b in [args.batch] + [x for x in (128, 1) if x < args.batch]:
            try:
                steps_b = args.steps if b > 1 else max(1, min(args.steps, 2))          # -b 1 is 512 decode steps per repetition
                pp_total, pp_samples = timed(lambda: pp_rep(b), steps_b, args.warmup if b > 1 else min(args.warmup, 1))
                row = dict(tok_s=round(steps_b * args.n_prompt / pp_total, 2), batch=b, steps=steps_b,
                           samples_tok_s=[round(args.n_prompt / s, 2) for s in pp_samples])
            except hip.Gl3Error as e:
                row = dict(batch=b, error=str(e))
            pp_rows.append(row)
        pp = pp_rows[0]

    # ---- llama-bench -d: tg behind an untimed prefill of d positions (token ids indexed by absolute position).  The KV rows below d
    # are written ONCE per depth (the reference re-runs the untimed prefill before every repetition; the timed window is the same)
    depth_rows = []
    for d in depths:
        try:
            t_pf = time.perf_counter()
            plan.prefill(toks[:d], 0, batch=args.batch if args.batch > 1 else None)
            torch.cuda.synchronize()
            t_pf = time.perf_counter() - t_pf

            def tg_at_depth(d=d):
                for i in range(args.n_gen):
                    plan.forward_decode(toks[d + i], d + i, copy=False)
            steps_d = max(1, min(args.steps, 3))
            tot, smp = timed(tg_at_depth, steps_d, 1)
            k = plan.profile_decode(toks[d + 64], d + 64)
            att = k.get("attention", {})
            kv_read = 2 * cfg.n_layers * (d + 64 + 1) * (cfg.kv_dim // world) * 4
            depth_rows.append(dict(depth=d, test="tg%d@d%d" % (args.n_gen, d), tok_s=round(steps_d * args.n_gen / tot, 2), steps=steps_d,
                                   samples_tok_s=[round(args.n_gen / x, 2) for x in smp], untimed_prefill_s=round(t_pf, 2),
                                   attention_us_per_layer=round(att["ms"] * 1e3 / cfg.n_layers, 2) if att.get("launches") else None,
                                   attention_launches_per_layer=(att["launches"] // cfg.n_layers) if att.get("launches") else None,
                                   kv_read_bytes_per_token=int(kv_read),
                                   kv_read_us_per_layer_at_hbm_peak=round(kv_read / cfg.n_layers / (HBM_PEAK_GBS * 1e9) * 1e6, 2)))
        except hip.Gl3Error as e:
            depth_rows.append(dict(depth=d, error=str(e)))
    if depths:
        plan.reset_kv()

    # ---- roofline of the dominant kernel: instrumented (eager, HIP events per launch) decode steps mid-sequence
    acc = None
    n_prof = 8
    for i in range(n_prof):
        k = plan.profile_decode(toks[64 + i], 64 + i)
        if acc is None:
            acc = k
        else:
            for name in k:
                for f in ("ms", "launches", "bytes"):
                    acc[name][f] += k[name][f]
    kern = {}
    for name, v in acc.items():
        if v["launches"]:
            us = v["ms"] / v["launches"] * 1e3
            kern[name] = dict(avg_us=round(us, 3), launches_per_token=v["launches"] // n_prof,
                              bytes_per_launch=v["bytes"] // v["launches"],
                              gbs=round(v["bytes"] / v["launches"] / (us * 1e-6) / 1e9, 1) if us > 0 else None)
    # dominant kernel (fused gate/up matvec): one HIP event pair around back-to-back launches over all layers' weights
    kclass = {}
    for name in ("matvec_qkv", "matvec_wo", "matvec_gateup", "matvec_down", "matvec_logits"):
        r = plan.profile_kernel(name, iters=20 if name != "matvec_logits" else 200)
        kclass[name] = dict(avg_us=round(r["avg_us"], 3), bytes_per_launch=r["bytes_per_launch"], gbs=round(r["gbs"], 1),
                            frac_of_hbm_peak=round(r["gbs"] / HBM_PEAK_GBS, 4))
    # `achieved` = the HIP-event-pair figure over back-to-back launches: it is the one the committed rocprofv3 kernel
    # statistics of this command agree with (the kernel inside the replayed decode graph).  The per-dispatch begin/end
    # timestamps of the instrumented EAGER steps are reported beside it; they run ~10 % longer (every dispatch carries two
    # event signals and starts cold behind an idle gap).
    dom = dict(kclass["matvec_gateup"])
    if "matvec_gateup" in kern and args.wtype == "q8_0":
        dom["avg_us_instrumented_steps"] = kern["matvec_gateup"]["avg_us"]
    kname = "matvec_q8t_kernel<PRO_RMS,EPI_SWIGLU> (fused RMSNorm + gate/up Q8_0 matvec + SwiGLU, %dx%d x2)" if args.wtype == "q8_0" else \
        "rmsnorm_f32_kernel + matvec_vl_kernel<WT_" + WT.split("_F32")[0] + ",EPI_SWIGLU> (gate/up Vector-API-order matvec + SwiGLU, %dx%d x2)"
    roofline = dict(bound="hbm", kernel=kname % (cfg.hidden // world, cfg.dim),
                    achieved=dom["gbs"], peak=HBM_PEAK_GBS, unit="GB/s", frac=round(dom["gbs"] / HBM_PEAK_GBS, 4),
                    traffic=None, avg_us=dom["avg_us"], avg_us_instrumented_steps=dom.get("avg_us_instrumented_steps"),
                    bytes_per_launch=dom["bytes_per_launch"],
                    method="avg_us: one HIP event pair on the plan's stream around 20 sweeps x %d layers of back-to-back launches of this "
                           "kernel over every layer's weights (includes the inter-kernel boundary; activations L2-warm); "
                           "avg_us_instrumented_steps: start/stop events passed into each dispatch (hipExtLaunchKernel), mean over %d "
                           "launches in %d eager decode steps at positions 64.." % (cfg.n_layers, n_prof * cfg.n_layers, n_prof))

    if args.wtype == "q4_0":
        # r6: the VALU floor of the Q4_0 gate/up matvec beside its HBM fraction.  Main loop of matvec_vl_kernel<WT_Q4_0, EPI_SWIGLU, RMS> (hipcc -S, one trip =
        # 64 (block, accumulator lane) units): 256 v_fma_mix_f32 + 128 v_pk_add_f16 + 128 v_and_or_b32 + 96 v_pk_add_f32 + 96 v_lshrrev_b32 + 32 v_pk_fma_f32 +
        # 64 v_cvt_f32_f16 + ~34 moves / address adds = 13.0 VALU instructions per unit; at the measured issue rates of those opcodes
        # (profiles/r03_valu_op_rates.txt, >= 2 wavefronts per SIMD: 2.04 / 2.09 / 2.08 / 2.15 / 1.17 / 2.15 / 1.94 / 1.2 ns) 25.1 ns per unit and SIMD.
        units = 2 * (cfg.hidden // world) * 8 * (cfg.dim // 32)
        floor_us = units / (1024 * 64) * 25.1e-3
        roofline["valu_floor"] = dict(valu_instructions_per_block_and_lane=13.0, ns_per_block_and_lane=25.1, floor_us=round(floor_us, 2),
                                      frac_of_valu_floor=round(floor_us / dom["avg_us"], 4),
                                      note="the nibble unpack + 4 rounded products + 3 adds + 1 fma per 32-element block and accumulator lane bound this kernel, not HBM")
    # measured-achievable peaks of THIS device beside the spec-sheet denominators; frac stays on the spec peak
    peaks = probe_peaks(local_rank) if rank == 0 else None
    if peaks:
        roofline["peak_measured"] = peaks["hbm_read_gbs"]
        roofline["frac_of_peak_measured"] = round(dom["gbs"] / peaks["hbm_read_gbs"], 4)
        roofline["peak_measured_detail"] = peaks

    # ---- batched prefill: the dominant GEMM (gate/up, int8 MFMA) and the other three, at the pp chunk size
    roofline_pp = None
    if pp is not None and "tok_s" in pp and args.wtype == "q8_0" and args.batch > 1:
        pk = {}
        for name in ("matvec_qkv", "matvec_wo", "matvec_gateup", "matvec_down"):
            r = plan.profile_prefill_kernel(name, min(args.batch, args.n_prompt), iters=3)
            pk[name.replace("matvec_", "gemm_")] = dict(avg_us=round(r["avg_us"], 2), int8_ops_per_launch=r["int8_ops_per_launch"], tops=round(r["tops"], 1),
                                                        frac_of_int8_mfma_peak=round(r["tops"] / INT8_MFMA_PEAK_TOPS, 4))
        g = pk["gemm_gateup"]
        ntok_pp = min(args.batch, args.n_prompt)
        # matrix-pipe time of the r4 kernel: three 32-cycle MFMAs per (32 x 32 tile, block) — the int8 dot and the two exact bf16-split
        # outer products s = wScale aScale, -B s (gl3_prefill_gemm2.h) — on 4 SIMDs x CUs; the honest MFMA-bound floor of this arithmetic
        tiles_blocks = 2 * (cfg.hidden // world // 32) * ((ntok_pp + 31) // 32) * (cfg.dim // 32)
        roofline_pp = dict(bound="mfma", kernel="pf_gemm3t_kernel (tall one-round tiling) / pf_gemm3_kernel (gate/up Q8_0 x int8-activation GEMM + SwiGLU, 2 x %dx%d x %d tokens)" %
                           (cfg.hidden // world, cfg.dim, ntok_pp),
                           achieved=g["tops"], peak=INT8_MFMA_PEAK_TOPS, unit="TOP/s", frac=g["frac_of_int8_mfma_peak"], traffic=None,
                           avg_us=g["avg_us"], int8_ops_per_launch=g["int8_ops_per_launch"],
                           dtype="i8 x i8 -> i32 MFMA + two bf16 MFMAs for the exact scale products; 2 f32 VALU ops per output and block (fma, add)",
                           mfma_instructions_per_launch=3 * tiles_blocks,
                           note="the int8-peak fraction counts only the int8 dot; the reference's per-32-block f32 arithmetic "
                                "(result += isum * (wScale * aScale), one rounding per operation) cannot reach 50 % of the int8 peak bit-exactly: r3 needed 4 VALU "
                                "lane-ops per output and block (VALU-bound), r4 moves two of them onto the matrix pipe as exact outer products, which "
                                "makes the pipe 3 x 32 cycles per tile-block = 1/3 int8 work at best; r6 (gl3_prefill_gemm3.h / gl3_prefill_gemm3t.h): mid-stage "
                                "barrier + partial vmcnt, scale-operand side table, chunk-major activations, one-round tall tiles; profiles/r06_prefill_gemm.md",
                           gemms=pk, method="one HIP event pair around 3 sweeps x %d layers per GEMM class" % cfg.n_layers)
        if peaks:
            roofline_pp["peak_measured"] = peaks["int8_mfma_tops"]
            roofline_pp["frac_of_peak_measured"] = round(g["tops"] / peaks["int8_mfma_tops"], 4)
            # time the matrix pipe needs for this launch's 3 MFMAs per tile-block at the measured int8 MFMA issue rate
            mfma_floor_us = 3 * tiles_blocks * 65536 / (peaks["int8_mfma_tops"] * 1e12) * 1e6
            roofline_pp["matrix_pipe_floor_us"] = round(mfma_floor_us, 1)
            roofline_pp["frac_of_matrix_pipe_floor"] = round(mfma_floor_us / g["avg_us"], 4)
            # r5: an MFMA and f32 VALU work of one SIMD do not execute concurrently on gfx950 (scripts/probes/mfma_overlap_probe.hip), so the
            # floor of this arithmetic is the SUM: 3 x 32 cycles of MFMA + 32 plain f32 operations at 2 cycles per tile-block
            # (profiles/r05_prefill_q8_bound.md)
            roofline_pp["mfma_plus_valu_floor_us"] = round(mfma_floor_us * (96 + 64) / 96, 1)
            roofline_pp["frac_of_mfma_plus_valu_floor"] = round(mfma_floor_us * (96 + 64) / 96 / g["avg_us"], 4)

    # HBM traffic of the dominant kernel: PMC counters cannot be read in-process (--pmc needs its own rocprofv3 pass).  The round's
    # separate FETCH_SIZE pass over this same command is committed as profiles/rNN_pmc_fetch_summary.csv (x2 gfx950 correction) together
    # with a .meta.json naming the SHA-256 of the kernel source it profiled.  When that hash equals the source of THIS build the counter
    # figure IS this kernel's traffic and fills `traffic` (the method string says where it comes from); otherwise `traffic` stays null and
    # the stale figure rides under a key that says so.
    try:
        import csv
        import glob
        import hashlib
        f = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_fetch_summary.csv")))[-1]
        # every file the dominant kernel and its launch geometry are built from (r5 hashed the kernel header alone: an edit to the launch code left the hash valid)
        src_sha = hashlib.sha256(b"".join(open(os.path.join(ROOT, "gpullama3.java_amd", "csrc", n), "rb").read()
                                          for n in ("gl3_decode_kernels.h", "gl3_seqsum.h", "gl3_ctx.h", "gl3_api.hip", "Makefile"))).hexdigest()
        meta_path = f.replace(".csv", ".meta.json")
        meta = json.load(open(meta_path)) if os.path.exists(meta_path) else {}
        for r in csv.reader(open(f)):
            if "matvec_q8t_kernel<0, 2" in r[0] and world == 1 and args.model == "llama-3-8b" and args.wtype == "q8_0":
                src = os.path.relpath(f, ROOT) + " (FETCH_SIZE x 2, separate rocprofv3 --pmc pass of this command)"
                if meta.get("kernel_sources_sha256") == src_sha:
                    roofline["traffic"] = int(r[-1])
                    roofline["method"] += "; traffic: not measured in this run — " + src + ", taken on this build's kernel source (sha256 " + src_sha[:12] + ")"
                else:
                    roofline["traffic_not_measured_in_this_run"] = dict(bytes_per_launch=int(r[-1]), source=src + "; kernel source changed since")
    except Exception:
        pass

    # whole-token algorithmic bytes: weights + norms + KV read/write + logits
    L, kvd = cfg.n_layers, cfg.kv_dim
    mat_elems = L * (cfg.q_dim * cfg.dim + 2 * kvd * cfg.dim + cfg.dim * cfg.q_dim + 3 * cfg.hidden * cfg.dim) + cfg.vocab * cfg.dim
    avg_pos = (args.n_gen - 1) / 2.0
    token_bytes = int(mat_elems * bpe) + (2 * L + 1) * cfg.dim * 4 + int(cfg.dim * bpe) + 2 * L * kvd * 4 * (avg_pos + 1) \
        + 2 * L * kvd * 4 + cfg.vocab * 4
    token_gbs = token_bytes * tg_tok_s / 1e9

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle_c
        # the oracle in the plan's mode: Vector-API dot order (256-bit species) for F16 / Q4_0 / f32-activation Q8_0
        o = oracle_c.COracle(model, vector_bits=0 if args.wtype == "q8_0" else 256, f32_activation=args.wtype == "q8_0_f32act")
        n_done, t1 = 0, time.perf_counter()
        while True:
            o.forward(toks[n_done], n_done)
            n_done += 1
            el = time.perf_counter() - t1
            if el > args.cpu_seconds or n_done >= args.n_gen:
                break
        cpu = dict(value=round(n_done / el, 4), unit="tok/s", cores=oracle_c.lib().orc_num_threads(), kind="port",
                   sample="tg%d at depth 0 (first %d decode steps of the same model/token stream, %.1f s)" % (n_done, n_done, el),
                   note="C restatement of forwardJava, rows / heads over an OpenMP pool; it quantises the activation once per matmul where the "
                        "Java path re-quantises per row, so it is FASTER than the JVM path it stands for (kind 'port', not 'reference')",
                   **host_cpu_info())
        # parity spot check on the full-size model, printed to stderr (the asserts live in tests/)
        ref = o.forward(toks[n_done], n_done)
        plan.reset_kv()
        for i in range(n_done):
            plan.forward_decode(toks[i], i, copy=False)
        got = plan.forward_decode(toks[n_done], n_done)
        err = float(np.max(np.abs(got - ref)) / np.max(np.abs(ref)))
        print("[bench] full-size parity vs CPU oracle at pos %d: rel err %.3e, argmax %d/%d" %
              (n_done, err, int(np.argmax(got)), int(np.argmax(ref))), file=sys.stderr)
        cpu["parity_rel_err_fullsize"] = err

    if rank == 0:
        mean = np.mean([args.n_gen / s for s in tg_samples])
        sd = float(np.std([args.n_gen / s for s in tg_samples], ddof=1)) if len(tg_samples) > 1 else 0.0
        out = {
            "metric": "tg128 tok/s (llama-bench), Llama-3-8B Q8_0" if (args.model == "llama-3-8b" and args.wtype == "q8_0" and args.n_gen == 128)
                      else "tg%d tok/s, %s %s" % (args.n_gen, args.model, WT),
            "value": round(tg_tok_s, 3), "unit": "tok/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(tg_total / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "i8" if args.wtype == "q8_0" else "f32", "data": "synthetic",
            "config": {"workload": "%s %s random-weight GGUF-layout model, tg%d at depth 0 (one step = %d decode tokens, logits D2H "
                                   "inside the timed region); pp%d -b %d reported beside it" %
                                   (cfg.name, WT, args.n_gen, args.n_gen, args.n_prompt, args.batch),
                       "parallelism": ("tp%d (row split, %s all-gathers)" % (world, "peer-write xGMI" if transport != "rccl" else "RCCL")) if world > 1 else "single GPU",
                       "ranks": world, "transport": (transport if world > 1 else None), "fold_mode": (list(plan.tp_fold_mode()) if world > 1 else None),
                       "ctx": cfg.ctx, "tokens": "java.util.Random(42)"},
            "tg_tok_s_mean": round(float(mean), 3), "tg_tok_s_stddev": round(sd, 3), "tg_samples_tok_s": [round(args.n_gen / s, 2) for s in tg_samples],
            "pp": pp, "pp_rows": pp_rows,
            **({"depth_rows": depth_rows} if depths else {}),
            "roofline": roofline, "roofline_pp": roofline_pp,
            "token_level": {"algorithmic_bytes_per_token": int(token_bytes), "achieved_gbs": round(token_gbs, 1),
                            "frac_of_hbm_peak": round(token_gbs / HBM_PEAK_GBS / max(world, 1), 4),
                            "roofline_tok_s": round(HBM_PEAK_GBS * 1e9 * world / token_bytes, 1)},
            "kernel_classes": kclass,
            "kernels_eager_events": kern,
            "cpu_baseline": cpu,
            "init": dict(plan.init_ms(), setup_s=round(setup_s, 2)),
        }
        print(json.dumps(out))
    plan.freeTornadoExecutionPlan()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def select_transport(want, world, rank, local_rank, dist, pkg, plan_mod, hip, torch):
    """Pick the tensor-parallel transport BEFORE the full-size plan is built.  The peer-write gather needs (1) every peer device
    addressable from this one (hipDeviceCanAccessPeer via gl3_tp_peer_access) and (2) hipIpcOpenMemHandle of a peer's uncached
    arena plus a gather that completes: a one-layer probe model runs two tensor-parallel decode steps end to end and the ranks
    compare their logits.  Any rank failing either check moves EVERY rank to RCCL.  With GL3_BENCH_SHARE_GPU (test hook: all ranks on
    device 0 of a one-GPU box) RCCL cannot be the fall-back — a communicator rejects duplicate devices — so a failed probe is fatal."""
    import ctypes as C
    import numpy as np
    if world == 1 or want == "rccl":
        return want
    share = bool(os.environ.get("GL3_BENCH_SHARE_GPU"))
    ok, why = 1, ""
    if not share:
        peers = (C.c_int32 * world)(*range(world))       # one rank per local device ordinal
        reach = C.c_int32()
        rc = hip.lib().gl3_tp_peer_access(local_rank, peers, world, C.byref(reach))
        if rc != 0 or reach.value != world:
            ok, why = 0, "hipDeviceCanAccessPeer: %d of %d peers reachable (rc %d)" % (reach.value, world, rc)

    exchanged = [False]

    def exchange(handle):
        # exactly ONE handle all-gather per rank: a rank whose plan fails BEFORE it gets here (gl3_create error, out of memory)
        # joins it afterwards with None, so the collectives of all ranks stay paired (r3 advisor finding: the failed rank used to
        # run ahead into the flags gather while its peers sat in this one)
        exchanged[0] = True
        out = [None] * world
        dist.all_gather_object(out, handle)
        if any(h is None for h in out):
            raise RuntimeError("a peer failed before the IPC handle exchange")
        return out

    flags = [None] * world
    dist.all_gather_object(flags, ok)
    if all(flags):
        synth = pkg.synth
        cfg = synth.ModelConfig("tp-probe", synth.ARCH_LLAMA, 512, 1024, 1, 16, 16, 32, 1024, 16, 1e-5, 10000.0, True)
        # every collective of the probe sits OUTSIDE the try blocks: a rank that fails still meets its peers at the next one
        probe, logits = None, None
        try:
            probe = plan_mod.HipMasterPlan(synth.make_numpy(cfg, seed=5), device=local_rank, tp_rank=rank, tp_size=world, p2p_exchange=exchange)
        except Exception as e:                          # noqa: BLE001
            ok, why = 0, "probe plan over IPC: %s" % e
        if not exchanged[0]:
            try:
                exchange(None)
            except RuntimeError:
                pass
        dist.all_gather_object(flags, ok)
        if all(flags):
            try:
                probe.forward_decode(3, 0)              # the gather kernel's spin is bounded: a dead link is an error, not a hang
                logits = probe.forward_decode(5, 1)
            except Exception as e:                      # noqa: BLE001
                ok, why = 0, "probe decode step: %s" % e
            sums = [None] * world
            dist.all_gather_object(sums, None if logits is None else logits.tobytes())
            if ok and any(x != sums[0] for x in sums):
                ok, why = 0, "probe logits differ between ranks"
            dist.all_gather_object(flags, ok)
        if probe is not None:
            probe.freeTornadoExecutionPlan()            # after the last collective: nobody unmaps an arena a peer still writes
    if all(flags):
        return want
    if not ok:
        print("rank %d: peer-write transport unavailable (%s)" % (rank, why), file=sys.stderr)
    if share:
        raise SystemExit("bench.py: peer-write transport failed with GL3_BENCH_SHARE_GPU set; RCCL cannot run several ranks on one device")
    return "rccl"


def self_spawn(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one process per GPU) through torch.distributed.run with
    this command line and pass rank 0's JSON line through.  Fails loudly when fewer than N devices are visible: a run must never
    print `n_gpus: 1` for `--gpus 8`."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if have < n and not os.environ.get("GL3_BENCH_SHARE_GPU"):
        raise SystemExit("bench.py: --gpus %d requested but only %d device(s) are visible (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES?)" % (n, have))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: hipIpcGetMemHandle of the peer-write arena needs it
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def bench_decode_batch(args, cfg, synth, plan_mod, pkg, torch, np, dev):
    """BASELINE configs[4]: B independent sequences advance one token per step through the int8-MFMA GEMM path
    (gl3_forward_decode_batch); one step reads every weight once, so the bound is still HBM.  Greedy ids are returned
    per step (B x 4 bytes D2H), logits stay on the device as with -Dllama.deviceSample."""
    B = args.decode_batch
    cfg = synth.ModelConfig(**{**cfg.__dict__, "ctx": args.n_gen + 8})
    # with a CPU baseline leg the host keeps the tensors (synth.make_torch) for the oracle; otherwise they stream through
    model = synth.make_torch(cfg, wtype=synth.GGML_Q8_0, seed=args.seed, device=dev) if not args.no_cpu_baseline else \
        synth.StreamModel(cfg, synth.GGML_Q8_0, synth.iter_torch(cfg, seed=args.seed, device=dev))
    t0 = time.time()
    plan = plan_mod.HipMasterPlan.initializeTornadoVMPlan(model, prefill_batch_size=B, n_seqs=B)
    setup_s = time.time() - t0
    toks = np.asarray(pkg.javarand.bench_tokens(cfg.vocab, args.n_gen * B), np.int32).reshape(args.n_gen, B)
    seqs = np.arange(B, dtype=np.int32)

    def rep():
        for i in range(args.n_gen):
            plan.forward_decode_batch(toks[i], seqs, np.full(B, i, np.int32), want_logits=False)

    for _ in range(args.warmup):
        rep()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        rep()
    torch.cuda.synchronize()
    total = time.perf_counter() - t1
    steps_s = args.steps * args.n_gen / total
    L, kvd = cfg.n_layers, cfg.kv_dim
    mat_elems = L * (cfg.q_dim * cfg.dim + 2 * kvd * cfg.dim + cfg.dim * cfg.q_dim + 3 * cfg.hidden * cfg.dim) + cfg.vocab * cfg.dim
    step_bytes = mat_elems * 34 // 32 + (2 * L + 1) * cfg.dim * 4
    gbs = step_bytes * steps_s / 1e9
    # CPU baseline beside it (kind "port"): B independent sequences on the C oracle are B single-sequence decodes — the reference's CPU path
    # has no batched decode — so a bounded sample of ONE sequence's steps gives the per-sequence rate; the host runs them back to back.
    cpu = None
    if not args.no_cpu_baseline:
        from oracle import oracle_c
        if getattr(model, "tensors", None):
            o = oracle_c.COracle(model)
            n_done, tc = 0, time.perf_counter()
            while True:
                o.forward(int(toks[n_done, 0]), n_done)
                n_done += 1
                el = time.perf_counter() - tc
                if el > args.cpu_seconds or n_done >= args.n_gen:
                    break
            cpu = dict(value=round(n_done / el, 4), unit="tok/s", cores=oracle_c.lib().orc_num_threads(), kind="port",
                       sample="%d single-sequence decode steps of sequence 0 (%.1f s): the CPU path has no batched decode, B sequences run one after the other "
                              "at this rate" % (n_done, el), **host_cpu_info())
    peaks = probe_peaks(0)
    roof_extra = dict(peak_measured=peaks["hbm_read_gbs"], frac_of_peak_measured=round(gbs / peaks["hbm_read_gbs"], 4), peak_measured_detail=peaks) if peaks else {}
    print(json.dumps({
        "metric": "static-batched decode B=%d tok/s, %s Q8_0" % (B, args.model), "value": round(steps_s * B, 2), "unit": "tok/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(total / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "i8", "data": "synthetic",
        "config": {"workload": "%s Q8_0 random weights, %d sequences x %d decode steps from position 0 (one bench step = %d batched "
                               "steps), greedy ids D2H" % (cfg.name, B, args.n_gen, args.n_gen), "parallelism": "single GPU"},
        "batched_steps_per_s": round(steps_s, 2), "ms_per_batched_step": round(1e3 / steps_s, 4),
        "roofline": {"bound": "hbm", "kernel": "whole batched step (weights read once per step)", "achieved": round(gbs, 1),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
                     "bytes_per_step": int(step_bytes), **roof_extra},
        "cpu_baseline": cpu,
        "init": dict(plan.init_ms(), setup_s=round(setup_s, 2))}))
    plan.freeTornadoExecutionPlan()


if __name__ == "__main__":
    main()
