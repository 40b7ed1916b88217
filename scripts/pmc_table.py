#!/usr/bin/env python3
"""Mean of every counter per (kernel, grid) from a rocprofv3 --pmc counter_collection.csv, as a small table on stdout / a CSV.
    python scripts/pmc_table.py <dir or csv> [name filter ...] [> out.csv]"""
import collections, csv, glob, os, sys
src, want = sys.argv[1], sys.argv[2:]
fs = [src] if os.path.isfile(src) else glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for f in fs:
    for r in csv.DictReader(open(f)):
        if want and not any(w in r["Kernel_Name"] for w in want):
            continue
        key = (r["Kernel_Name"][:110], r["Grid_Size"])
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[key] = (r["Workgroup_Size"], r["VGPR_Count"], r.get("LDS_Block_Size", ""))
counters = sorted({c for v in acc.values() for c in v})
w = csv.writer(sys.stdout)
w.writerow(["Kernel_Name", "Grid_Size", "Workgroup_Size", "VGPR_Count", "LDS_Block_Size", "Dispatches"] + counters)
for key, v in sorted(acc.items()):
    w.writerow([key[0], key[1], *meta[key], max(len(x) for x in v.values())] + [round(sum(v[c]) / len(v[c]), 1) if c in v else "" for c in counters])
