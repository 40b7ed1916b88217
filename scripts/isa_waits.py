"""Per-kernel ISA summary (loads, s_waitcnt vmcnt(0) vs partial waits, scratch, VGPRs) of a hipcc -S listing:
python scripts/isa_waits.py file.s  — a high vmcnt(0) share on a kernel with a software-pipelined load ring means the
compiler serialised the ring ("conditional loads")."""
import re, sys
cur = None; st = {}
for line in open(sys.argv[1]):
    m = re.match(r"^(_Z\w+):", line)
    if m: cur = m.group(1); st[cur] = dict(ld=0, w0=0, wp=0, scr=0, mfma=0, vgpr=0); continue
    if line.startswith(".Lfunc_end"): cur = None; continue
    m = re.match(r"\s*\.set (_Z\w+)\.num_vgpr, (\d+)", line)
    if m and m.group(1) in st: st[m.group(1)]["vgpr"] = int(m.group(2))
    if cur is None: continue
    s = st[cur]
    if "global_load" in line or "buffer_load" in line: s["ld"] += 1
    if "scratch_" in line: s["scr"] += 1
    if "mfma" in line: s["mfma"] += 1
    m = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", line)
    if m: s["w0" if m.group(1) == "0" else "wp"] += 1
print("%-90s %5s %5s %5s %5s %5s %5s" % ("kernel", "loads", "vm0", "vmN", "scr", "mfma", "vgpr"))
for k, s in st.items():
    if s["ld"]: print("%-90s %5d %5d %5d %5d %5d %5d" % (k[:90], s["ld"], s["w0"], s["wp"], s["scr"], s["mfma"], s["vgpr"]))
