"""How busy is the device over the bench's timed steps?  Reads a rocprofv3 kernel trace (csv) and reports, for the middle of the trace (setup and the per-kernel
roofline pass cut off), the share of the time with 0 / 1 / 2 / 3+ kernels in flight, and which kernels run alone for the longest.
usage: trace_gaps.py <kernel_trace.csv> [from_quantile to_quantile of the extension kernels' start times]"""
import csv, sys, re, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
lo_f = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
hi_f = float(sys.argv[3]) if len(sys.argv) > 3 else 0.85
T0, T1 = rows[0][0], max(r[1] for r in rows)
# the timed steps: where the extension kernels are (the fractions are quantiles of THEIR start times)
xs = sorted(s for s, e, n in rows if "extpk_kernel" in n)
a, b = xs[int(len(xs) * lo_f)], xs[int(len(xs) * hi_f)]
ev = []
for s, e, n in rows:
    if e <= a or s >= b: continue
    s, e = max(s, a), min(e, b)
    short = re.sub(r"\(.*", "", n)[:48]
    ev.append((s, 1, short)); ev.append((e, -1, short))
ev.sort(key=lambda t: (t[0], t[1]))
hist = collections.Counter(); alone = collections.Counter()
live = collections.Counter(); cur = 0; last = a
for t, d, n in ev:
    dt = t - last
    if dt > 0:
        hist[min(cur, 4)] += dt
        if cur == 1:
            alone[next(k for k, v in live.items() if v > 0)] += dt
    last = t
    cur += d; live[n] += d
hist[min(cur, 4)] += b - last
tot = b - a
print("window %.1f ms of the trace's %.1f ms" % (tot / 1e6, (T1 - T0) / 1e6))
for k in range(5):
    print("  %s kernels in flight: %5.1f %%" % (str(k) if k < 4 else "4+", 100.0 * hist[k] / tot))
print("alone for the longest:")
for n, v in alone.most_common(12):
    print("  %-50s %6.2f %%" % (n, 100.0 * v / tot))
