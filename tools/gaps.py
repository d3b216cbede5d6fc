"""GPU-idle gaps of the last training step in a three-stream kernel trace, grouped by the kernels around them:  python tools/gaps.py <t_kernel_trace.csv>"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Stream_Id"]) for r in rows if "rocclr_copy" not in r["Kernel_Name"])
ad = [e for e in ev if "k_adamw" in e[2]]
t0, t1 = ad[-2][1], ad[-1][1]
step = [e for e in ev if e[0] >= t0 and e[1] <= t1 + 1]
short = lambda n: (re.search(r"(k_[a-z0-9_]+)", n) or re.match(r"(.{0,30})", n)).group(1)
cur, prev, gaps = t0, "start", collections.defaultdict(lambda: [0, 0])
for s, e, n, q in step:
    if s > cur + 2000:
        g = gaps[(prev, short(n))]
        g[0] += s - cur
        g[1] += 1
    if e > cur:
        cur, prev = e, short(n)
print(f"step {(t1 - t0) / 1e6:.2f} ms, idle in gaps > 2 us: {sum(v[0] for v in gaps.values()) / 1e6:.2f} ms")
for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:10]:
    print(f"  {v[0] / 1e3:7.1f} us  n {v[1]:3d}  avg {v[0] / v[1] / 1e3:5.1f} us   {k[0]} -> {k[1]}")
