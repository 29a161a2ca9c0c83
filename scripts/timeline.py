#!/usr/bin/env python3
"""Timeline of a rocprofv3 --kernel-trace run of bench.py: per 1-ms bin of the last steps, how many kernels of each family were
running (seed / chain / ext / other) and which fraction of the bin had no kernel at all.  usage: timeline.py <trace.csv> [bins=120]"""
import csv, sys, collections
p = sys.argv[1]; nb = int(sys.argv[2]) if len(sys.argv) > 2 else 120
rows = list(csv.DictReader(open(p)))
ev = []
for r in rows:
    n = r["Kernel_Name"]; s = int(r["Start_Timestamp"]); e = int(r["End_Timestamp"])
    fam = "ext" if ("ext" in n and "chain" not in n) else "chain" if ("chain_" in n or "emit" in n or "merge" in n) else \
          "seed" if any(k in n for k in ("smem_", "locate", "expand", "cand_", "pack_reads", "filter")) else "other"
    ev.append((s, e, fam, n))
ev.sort()
t1 = max(e for _, e, _, _ in ev); t0 = t1 - nb * 1_000_000
print("bin(ms)  idle%   seed chain ext other  (average number of kernels running)")
for b in range(nb):
    lo = t0 + b * 1_000_000; hi = lo + 1_000_000
    cov = collections.Counter(); ivs = []
    for s, e, fam, n in ev:
        if e <= lo or s >= hi: continue
        a, z = max(s, lo), min(e, hi); cov[fam] += z - a; ivs.append((a, z))
    ivs.sort(); busy = 0; cur = lo
    for a, z in ivs:
        if z > cur: busy += z - max(a, cur); cur = z
    print(f"{b:4d}   {100 * (1 - busy / 1e6):5.1f}   " + " ".join(f"{cov[f] / 1e6:5.2f}" for f in ("seed", "chain", "ext", "other")))
