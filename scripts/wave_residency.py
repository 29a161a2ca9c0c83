#!/usr/bin/env python3
"""Who is resident where: the records of bmh_wtrace_start / _stop (csrc/wtrace.h: one per wave of the seeding, chaining and packed
extension kernels -- kernel, HW_ID, XCC_ID, start / end in 100 MHz ticks) turned into waves of each kernel family per SIMD over time.

    from wave_residency import trace, summary
    with trace(L) as t:            # L = the loaded library
        ... launch work, synchronise ...
    print(summary(t.records))

What the summary answers: while the extension of one batch runs, how many waves of the other batch's
gather-bound kernels share its SIMDs -- per-SIMD residency sampled at `samples` moments of the traced interval."""
import ctypes as C
import numpy as np

REC = np.dtype([("kid", "<u4"), ("hw", "<u4"), ("xcc", "<u4"), ("aux", "<u4"), ("t0", "<u8"), ("t1", "<u8")])
NAMES = {1: "smem_forward", 2: "smem_backward", 3: "locate", 4: "expand", 5: "smem_filter", 6: "pack_reads", 7: "cand_scatter",
         16: "ext_closed_form", 17: "extpk<G,P>", 18: "extpk_persist", 19: "extend16/wide",
         32: "chain_classify", 33: "chain_lane", 34: "chain_lane_list", 35: "chain_wave", 36: "emit", 37: "merge", 38: "chain_sub"}
FAMILIES = ("seeding", "extension", "chaining")


def family(kid):
    return np.where(kid < 16, 0, np.where(kid < 32, 1, 2))


class trace:
    def __init__(self, L, cap=6_000_000):
        self.L, self.cap, self.records, self.reported = L, cap, None, 0

    def __enter__(self):
        rc = self.L.bmh_wtrace_start(self.cap)
        if rc != 0:
            raise RuntimeError("bmh_wtrace_start: %d %s" % (rc, self.L.bmh_last_error()))
        return self

    def __exit__(self, *exc):
        buf = np.zeros(self.cap, REC)
        n = self.L.bmh_wtrace_stop(buf.ctypes.data_as(C.c_void_p), self.cap)
        self.reported = int(n)
        self.records = buf[: int(self.L.bmh_wtrace_kept())]
        return False


def simd_key(r):
    """(XCD, shader engine, shader array, CU, SIMD) of a record as one integer.  HW_ID (gfx9 layout): wave 3:0, SIMD 5:4, pipe 7:6, CU 11:8, SH 12, SE 15:13"""
    hw = r["hw"].astype(np.int64)
    return ((r["xcc"].astype(np.int64) & 0xF) << 12) | (((hw >> 13) & 7) << 9) | (((hw >> 12) & 1) << 8) | (((hw >> 8) & 0xF) << 4) | ((hw >> 4) & 3)


def summary(recs, samples=400, title=""):
    if len(recs) == 0:
        return "(no wave records)"
    out = []
    t_lo, t_hi = int(recs["t0"].min()), int(recs["t1"].max())
    span_ms = (t_hi - t_lo) / 1e5
    keys, simd = np.unique(simd_key(recs), return_inverse=True)
    n_simd = len(keys)
    fam = family(recs["kid"])
    out.append(f"{title}{len(recs)} waves on {n_simd} SIMDs over {span_ms:.2f} ms")
    # per kernel: waves, mean lifetime, wave-milliseconds, mean resident waves per SIMD over the interval
    out.append(f"  {'kernel':18s} {'waves':>9s} {'mean life us':>13s} {'wave-ms':>10s} {'resident / SIMD':>16s} {'first wave at ms':>17s} {'10% / 50% / 90% of its waves started by ms':>44s} {'last ends':>10s}")
    life = (recs["t1"] - recs["t0"]).astype(np.float64) / 100.0        # us
    for kid in np.unique(recs["kid"]):
        m = recs["kid"] == kid
        st = np.sort(recs["t0"][m]).astype(np.float64)
        q = [(st[min(len(st) - 1, int(len(st) * f))] - t_lo) / 1e5 for f in (0.1, 0.5, 0.9)]
        out.append(f"  {NAMES.get(int(kid), str(int(kid))):18s} {int(m.sum()):9d} {life[m].mean():13.1f} {life[m].sum() / 1e3:10.1f} {life[m].sum() / 1e3 / span_ms / n_simd:16.2f} "
                   f"{(st[0] - t_lo) / 1e5:17.2f} {q[0]:14.2f} / {q[1]:6.2f} / {q[2]:6.2f} {'':14s} {(int(recs['t1'][m].max()) - t_lo) / 1e5:10.2f}")
    # sampled residency: waves of each family on every SIMD at `samples` moments
    ts = np.linspace(t_lo, t_hi, samples + 2)[1:-1]
    acc = np.zeros((len(FAMILIES), 9))                      # histogram of resident waves per SIMD (0..8+) per family
    both = np.zeros((4,))                                   # SIMD-samples: [ext only, ext + seeding/chaining, seeding/chaining only, empty]
    co_hist = {}
    t0, t1 = recs["t0"], recs["t1"]
    for t in ts:
        m = (t0 <= t) & (t1 > t)
        cnt = np.zeros((n_simd, len(FAMILIES)), np.int64)
        np.add.at(cnt, (simd[m], fam[m]), 1)
        for f in range(len(FAMILIES)):
            acc[f] += np.bincount(np.minimum(cnt[:, f], 8), minlength=9)
        e, o = cnt[:, 1] > 0, (cnt[:, 0] + cnt[:, 2]) > 0
        both += [np.sum(e & ~o), np.sum(e & o), np.sum(~e & o), np.sum(~e & ~o)]
        for k, v in zip(*np.unique(cnt[:, 1] * 16 + np.minimum(cnt[:, 0] + cnt[:, 2], 15), return_counts=True)):
            co_hist[int(k)] = co_hist.get(int(k), 0) + int(v)
    tot = both.sum()
    out.append(f"  SIMD-samples ({samples} moments x {n_simd} SIMDs): extension alone {both[0] / tot:.1%}, extension + seeding/chaining waves side by side {both[1] / tot:.1%}, "
               f"seeding/chaining alone {both[2] / tot:.1%}, none of the traced kernels {both[3] / tot:.1%}")
    for f, nm in enumerate(FAMILIES):
        h = acc[f] / acc[f].sum()
        out.append(f"  resident {nm:9s} waves per SIMD: " + " ".join(f"{k}{'+' if k == 8 else ''}:{h[k]:.1%}" for k in range(9)) + f"  mean {(h * np.arange(9)).sum():.2f}")
    top = sorted(co_hist.items(), key=lambda kv: -kv[1])[:10]
    out.append("  most frequent (extension waves, other waves) per SIMD: " + ", ".join(f"({k >> 4},{k & 15}{'+' if (k & 15) == 15 else ''}) {v / tot:.1%}" for k, v in top))
    return "\n".join(out)
