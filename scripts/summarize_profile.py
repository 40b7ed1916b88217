#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/prof_rNN/{trace,pmc_fetch}) into small files under profiles/.

    python scripts/summarize_profile.py gpurun_out/prof_r1 r01
"""
import collections
import csv
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
os.makedirs(out, exist_ok=True)

import glob
rows = list(csv.DictReader(open(glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)[0])))
with open(os.path.join(out, tag + "_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "StdDev"])
    for r in rows:
        if "gl3" in r["Name"] or "pf_" in r["Name"]:
            w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["StdDev"]])

pmc_dir = os.path.join(src, "pmc_fetch")
if os.path.isdir(pmc_dir):
    f = glob.glob(os.path.join(pmc_dir, "**", "*counter_collection.csv"), recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "gl3" in r["Kernel_Name"] or "pf_" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"], r["Counter_Name"], r["Grid_Size"], r["LDS_Block_Size"], r["VGPR_Count"])].append(float(r["Counter_Value"]))
    with open(os.path.join(out, tag + "_pmc_fetch_summary.csv"), "w", newline="") as fo:
        w = csv.writer(fo)
        w.writerow(["Kernel_Name", "Counter", "Grid_Size", "LDS_Block_Size", "VGPR_Count", "Dispatches", "Mean_Counter_Value",
                    "HBM_read_bytes_per_launch (= FETCH_SIZE KiB-units x 1024 x 2, gfx950 half-count correction of MI355X_MICROARCH.md)"])
        for k, v in sorted(acc.items()):
            mean = sum(v) / len(v)
            w.writerow(list(k) + [len(v), round(mean, 3), int(mean * 1024 * 2)])
    # which kernel source the counters belong to: bench.py fills roofline.traffic from this table only while the hash still matches
    import hashlib
    import json
    csrc = os.path.join(os.path.dirname(out), "gpullama3.java_amd", "csrc")
    files = ("gl3_decode_kernels.h", "gl3_seqsum.h", "gl3_ctx.h", "gl3_api.hip", "Makefile")      # the dominant decode kernel + its launch geometry + build flags (bench.py hashes the same list)
    json.dump({"kernel_sources_sha256": hashlib.sha256(b"".join(open(os.path.join(csrc, n), "rb").read() for n in files)).hexdigest(), "files": list(files),
               "command": "rocprofv3 --pmc FETCH_SIZE --output-format csv -- python bench.py --steps 1 --no-pp --no-cpu-baseline"},
              open(os.path.join(out, tag + "_pmc_fetch_summary.meta.json"), "w"))
print("wrote", [x for x in os.listdir(out) if x.startswith(tag)])

# ---- round 3 additions: SQ / matrix-pipe counters of the batched-prefill kernels, their FETCH_SIZE, further kernel statistics
def counter_table(sub, out_name, want):
    d = os.path.join(src, sub)
    if not os.path.isdir(d):
        return
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for r in csv.DictReader(open(f)):
        if any(w in r["Kernel_Name"] for w in want):
            key = (r["Kernel_Name"], r["Grid_Size"])
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[key] = (r["Workgroup_Size"], r["VGPR_Count"])
    counters = sorted({c for v in acc.values() for c in v})
    with open(os.path.join(out, out_name), "w", newline="") as fo:
        w = csv.writer(fo)
        extra = []
        if "SQ_VALU_MFMA_BUSY_CYCLES" in counters and "SQ_BUSY_CYCLES" in counters:
            extra = ["MfmaUtil_% (= SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs), kernel cycles = SQ_BUSY_CYCLES / 32 SQs)",
                     "VALU_insts_per_MFMA", "VALU_active_% of wave cycles", "issue_stall_% of wave cycles", "parked_% of wave cycles"]
        w.writerow(["Kernel_Name", "Grid_Size", "Workgroup_Size", "VGPR_Count", "Dispatches"] + counters + extra)
        for key, v in sorted(acc.items()):
            mean = {c: sum(x) / len(x) for c, x in v.items()}
            n = max(len(x) for x in v.values())
            row = [key[0], key[1], meta[key][0], meta[key][1], n] + [round(mean.get(c, 0.0), 1) for c in counters]
            if extra:
                busy = mean.get("SQ_BUSY_CYCLES", 0.0)
                kern_cycles = busy / 32.0 if busy else 0.0
                mf = mean.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
                wc = mean.get("SQ_WAVE_CYCLES", 0.0)
                row += [round(100.0 * mf / (kern_cycles * 1024.0), 2) if kern_cycles else "",
                        round(mean.get("SQ_INSTS_VALU", 0.0) / mean["SQ_INSTS_MFMA"], 1) if mean.get("SQ_INSTS_MFMA") else "",
                        round(100.0 * mean.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, 1) if wc else "",
                        round(100.0 * mean.get("SQ_WAIT_INST_ANY", 0.0) / wc, 1) if wc else "",
                        round(100.0 * mean.get("SQ_WAIT_ANY", 0.0) / wc, 1) if wc else ""]
            w.writerow(row)


counter_table("pmc_pp_q8", tag + "_pmc_prefill_q8_mfma_valu.csv", ["pf_gemm", "pf_scores", "pf_pv", "pf_softmax", "pf_attn"])
counter_table("pmc_pp_f16", tag + "_pmc_prefill_f16_mfma_valu.csv", ["gemm_f16", "gemm_vlq"])
counter_table("pmc_pp_q8_fetch", tag + "_pmc_prefill_q8_fetch.csv", ["pf_gemm"])


def stats_table(sub, out_name, want):
    d = os.path.join(src, sub)
    if not os.path.isdir(d):
        return
    f = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)[0]
    with open(os.path.join(out, out_name), "w", newline="") as fo:
        w = csv.writer(fo)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "StdDev"])
        for r in csv.DictReader(open(f)):
            if any(x in r["Name"] for x in want):
                w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["StdDev"]])


stats_table("trace_f16", tag + "_f16_prefill_kernel_stats.csv", ["gl3", "pf_"])
stats_table("trace_q4", tag + "_q4_0_prefill_kernel_stats.csv", ["gl3", "pf_"])
stats_table("trace_bd", tag + "_bd32_kernel_stats.csv", ["gl3", "pf_", "bdw"])
import shutil
if os.path.exists(os.path.join(src, "valu_rate_probe.txt")):
    shutil.copy(os.path.join(src, "valu_rate_probe.txt"), os.path.join(out, tag + "_valu_rate_probe.txt"))
print("wrote", sorted(x for x in os.listdir(out) if x.startswith(tag)))
