#!/usr/bin/env python3
"""VGPRs / SGPRs / scratch / static LDS of every kernel of the library, from `hipcc -S` listings of the four device translation
units (no GPU needed).  A kernel with scratch bytes has spills: look at it with scripts/isa_waits.py.
    python scripts/kernel_resources.py profiles/rNN_kernel_resources.csv"""
import csv, os, subprocess, sys, tempfile

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpullama3.java_amd", "csrc")
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "profiles", "kernel_resources.csv")
flags = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden --cuda-device-only -S".split()
rows = []
with tempfile.TemporaryDirectory() as tmp:
    procs = []
    for f in ("gl3_api", "gl3_prefill", "gl3_sample", "gl3_tp"):
        s = os.path.join(tmp, f + ".s")
        procs.append((f, s, subprocess.Popen(["/opt/rocm/bin/hipcc", *flags, "-o", s, f + ".hip"], cwd=src, stderr=subprocess.DEVNULL)))
    for f, s, p in procs:
        if p.wait() != 0:
            raise SystemExit("hipcc failed on " + f)
        k, d = None, {}
        for ln in open(s):
            t = ln.split()
            if not t:
                continue
            if t[0] == ".amdhsa_kernel":
                k, d = t[1], {}
            elif t[0].startswith(".amdhsa_") and len(t) > 1:
                d[t[0]] = t[1]
            elif t[0] == ".end_amdhsa_kernel" and k:
                name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
                rows.append([f + ".hip", name, d.get(".amdhsa_next_free_vgpr"), d.get(".amdhsa_next_free_sgpr"),
                             d.get(".amdhsa_private_segment_fixed_size"), d.get(".amdhsa_group_segment_fixed_size")])
                k = None
with open(out, "w", newline="") as fo:
    w = csv.writer(fo)
    w.writerow(["translation_unit", "kernel", "vgprs (incl. AGPRs)", "sgprs", "scratch_bytes_per_lane", "static_lds_bytes"])
    for r in sorted(rows):
        w.writerow(r)
spill = [r for r in sorted(rows) if r[4] not in ("0", None)]
print(len(rows), "kernels,", len(spill), "with scratch")
for r in spill:
    print("  %-120s vgprs %s scratch %s" % (r[1][:120], r[2], r[4]))
