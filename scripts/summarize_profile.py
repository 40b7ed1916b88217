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
print("wrote", [x for x in os.listdir(out) if x.startswith(tag)])
