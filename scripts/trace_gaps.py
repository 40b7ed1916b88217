"""Kernel-time vs wall-time per token from a rocprofv3 --kernel-trace CSV: python scripts/trace_gaps.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
ks.sort()
# tokens start at embed kernels
idx = [i for i, k in enumerate(ks) if "embed" in k[2]]
idx = idx[len(idx) // 2:]          # second half = the timed loop
tot_wall = tot_kern = 0
per = collections.defaultdict(lambda: [0, 0])
gaps = []
for a, b in zip(idx[:-1], idx[1:]):
    seg = ks[a:b]
    tot_wall += ks[b][0] - seg[0][0]
    tot_kern += sum(e - s for s, e, _ in seg)
    for (s, e, nm), nxt in zip(seg, seg[1:] + [ks[b]]):
        per[nm[:60]][0] += e - s; per[nm[:60]][1] += 1
        gaps.append(nxt[0] - e)
n = len(idx) - 1
print("tokens %d: wall %.1f us/token, kernels %.1f us/token, gaps %.1f us/token (%d launches/token, median gap %.2f us)" %
      (n, tot_wall / n / 1e3, tot_kern / n / 1e3, (tot_wall - tot_kern) / n / 1e3, len(gaps) // n, sorted(gaps)[len(gaps) // 2] / 1e3))
for nm, (t, c) in sorted(per.items(), key=lambda kv: -kv[1][0]):
    print("  %-62s %7.1f us/token  %5.2f us avg x %d" % (nm, t / n / 1e3, t / c / 1e3, c // n))
