"""Concurrency of a three-stream step from a rocprofv3 kernel trace: how long 0 / 1 / 2 / 3+ kernels are in flight, and which kernels run ALONE.
    python tools/overlap.py <..._kernel_trace.csv> [steps=3]
The trace of tools/train_once.py holds `steps` training steps after the constructor; the last step is analysed (from its first k_prologue_fwd to its k_adamw)."""
import csv, sys, collections, re
path = sys.argv[1]
rows = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"])) for r in csv.DictReader(open(path))]
rows.sort(key=lambda r: r[1])
starts = [i for i, r in enumerate(rows) if "k_prologue_fwd" in r[0]]
ends = [i for i, r in enumerate(rows) if "k_adamw" in r[0]]
a, b = starts[-1], ends[-1]
step = rows[a:b + 1]
t0, t1 = step[0][1], max(r[2] for r in step)
ev = []
for n, s, e, q in step:
    ev.append((s, 1, n)); ev.append((e, -1, n))
ev.sort()
live = collections.Counter(); depth_time = collections.Counter(); alone = collections.Counter(); last = t0; cur = 0
def short(n):
    m = re.search(r"(k_[a-z0-9_]+)", n)
    return m.group(1) if m else n[:30]
for t, d, n in ev:
    if t > last:
        depth_time[min(cur, 3)] += t - last
        if cur == 1:
            alone[short(next(k for k, v in live.items() if v > 0))] += t - last
    last = t
    live[n] += d; cur += d
tot = t1 - t0
print(f"step {tot / 1e6:.2f} ms, {len(step)} launches; sum of kernel durations {sum(e - s for _, s, e, _ in step) / 1e6:.2f} ms")
for k in range(4): print(f"  {k}{'+' if k == 3 else ' '} kernels in flight: {depth_time[k] / 1e6:6.2f} ms ({100 * depth_time[k] / tot:4.1f} %)")
print("  alone on the chip, by kernel (ms):", ", ".join(f"{k} {v / 1e6:.2f}" for k, v in alone.most_common(12)))
