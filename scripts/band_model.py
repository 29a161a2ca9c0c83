"""(CPU) What the extension kernels must compute vs what they do compute, on the real jobs of the bench workload.

usage: python scripts/band_model.py gpurun_out/jobs150.npz
Input: the job arrays dumped by scripts/band_probe.py on the GPU box.  Runs the checker's per-row trace of ksw_extend2 (trimmed
range [beg,end), potential bound, running max / gscore) and prints: reference cells, the cells the packed kernels compute under
the stop rules (none / potential over all cells incl. zero ones = round 2 / potential over the non-zero frontier = round 3 /
the same + the out3-only rule), by query-length class.
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import oracle_py

z = np.load(sys.argv[1])
q, qoff, qlen, t, toff, tlen, h0 = (np.ascontiguousarray(z[k]) for k in ("q", "qoff", "qlen", "t", "toff", "tlen", "h0"))
nj = len(qlen)
orc = oracle_py.Oracle(); lib = orc.lib
_u8p, _u32p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32)
lib.oracle_extend_trace.restype = C.c_uint64
lib.oracle_extend_trace.argtypes = [C.c_uint32, _u8p, _u32p, _u32p, _u8p, _u32p, _u32p, _u32p, C.POINTER(oracle_py.KswParams), _u32p, C.POINTER(C.c_int16), C.c_uint64]
cap = int(tlen.astype(np.int64).sum()) * 5 + 16
trace = np.zeros(cap, np.int16); rows = np.zeros(nj, np.uint32)
p = oracle_py.default_params()
used = lib.oracle_extend_trace(nj, q.ctypes.data_as(_u8p), qoff.ctypes.data_as(_u32p), qlen.ctypes.data_as(_u32p), t.ctypes.data_as(_u8p), toff.ctypes.data_as(_u32p),
                               tlen.ctypes.data_as(_u32p), h0.ctypes.data_as(_u32p), C.byref(p), rows.ctypes.data_as(_u32p), trace.ctypes.data_as(C.POINTER(C.c_int16)), cap)
T = trace[:used].reshape(-1, 5).astype(np.int64)
ql, tl, hh, rows = qlen.astype(np.int64), tlen.astype(np.int64), h0.astype(np.int64), rows.astype(np.int64)
off = np.concatenate([[0], np.cumsum(rows)])
jobid = np.repeat(np.arange(nj), rows)
rowidx = np.arange(len(T)) - off[jobid]
beg, end, U, MX, GS = T.T
width = (end - beg).clip(0)
# closed-form proxy: <= 2 mismatches on the main diagonal, no N, tlen >= qlen
mm = np.zeros(nj, np.int64)
for i in range(nj):
    n = min(int(ql[i]), int(tl[i]))
    a = q[qoff[i]: qoff[i] + n]; b = t[toff[i]: toff[i] + n]
    mm[i] = int(((a != b) | (a > 3) | (b > 3)).sum())
cf = (mm <= 2) & (tl >= ql)
dp = ~cf


def cls_cols(x):
    return np.where(x <= 128, np.maximum(2, (x + 15) // 16) * 16, np.where(x <= 136, 136, np.where(x <= 144, 144, np.where(x <= 160, 160, np.where(x <= 192, 192, np.where(x <= 224, 224, np.where(x <= 256, 256, 288)))))))


cc = cls_cols(ql)
EB = p.end_bonus


def first_row(flag):
    r = np.full(nj, 10 ** 9)
    np.minimum.at(r, jobid[flag], rowidx[flag] + 1)
    return np.minimum(rows, r)


# the round-2 rule let zero cells carry potential: column 0 alone gave U >= a * (qlen - 1) in every row
U_r2 = np.maximum(U, p.a * (ql[jobid] - 1))
rules = {
    "no early stop": rows,
    "round 2 (zero cells counted)": first_row((U_r2 <= MX) & (U_r2 < GS)),
    "non-zero frontier, raw-exact": first_row((U <= MX) & (U < GS)),
    "non-zero frontier + out3 rule": first_row((U <= MX) & ((U < GS) | ((U <= MX - EB) & (GS <= MX - EB)))),
}
print(f"{nj} jobs, {dp.sum()} to the DP (closed-form proxy {cf.mean():.3f}); reference cells {width.sum()} ({width[dp[jobid]].sum()} in DP jobs)")
base = None
for name, kr in rules.items():
    comp = (kr * cc)[dp].sum()
    inr = (rowidx < kr[jobid]) & dp[jobid]
    band = (width * inr).sum()
    base = base or comp
    print(f"  {name:58s}: rows {kr[dp].sum():10d} computed cells {comp:12d} ({comp / base:.3f}) band share {band / comp:.3f}  computed / reference(all) {comp / width.sum():.3f}")
kr = rules["non-zero frontier + out3 rule"]
print("by class, last rule:")
for lo, hi in ((1, 64), (65, 96), (97, 112), (113, 128), (129, 136), (137, 192), (193, 288)):
    m = dp & (ql >= lo) & (ql <= hi)
    if not m.any():
        continue
    print(f"  qlen {lo:3d}-{hi:3d}: jobs {m.sum():6d} rows old {rows[m].sum():9d} new {kr[m].sum():9d} share of computed {(kr * cc)[m].sum() / (kr * cc)[dp].sum():.3f}")
print("by diagonal mismatches (DP jobs, last rule):")
tot = (kr * cc)[dp].sum()
for lo, hi in ((3, 3), (4, 4), (5, 6), (7, 9), (10, 19), (20, 39), (40, 999)):
    m = dp & (mm >= lo) & (mm <= hi)
    print(f"  mm {lo:3d}-{hi:3d}: jobs {m.sum():6d} share of computed {(kr * cc)[m].sum() / tot:.3f} mean rows {kr[m].mean():.0f} mean qlen {ql[m].mean():.0f} reached end (gscore>0) {(np.maximum.reduceat(GS, off[:-1][rows>0])[m[rows>0]] > 0).mean() if m.any() else 0:.2f}")
m = dp & ~((mm <= 2)) 
print("  tlen < qlen jobs among DP:", (dp & (tl < ql)).sum())

# ---- a narrow sliding window for the flanks that look unrelated: W columns as four lane blocks that follow `beg`; a job whose band
# outgrows the window is redone in its full class.  Routed by the mismatches among the first 32 diagonal columns (the prefilter sees them).
if len(sys.argv) > 2 and sys.argv[2] == "window":
    kr = rules["non-zero frontier + out3 rule"]
    mm32 = np.zeros(nj, np.int64)
    for i in range(nj):
        n = min(int(ql[i]), int(tl[i]), 32)
        a = q[qoff[i]: qoff[i] + n]; b = t[toff[i]: toff[i] + n]
        mm32[i] = int(((a != b) | (a > 3) | (b > 3)).sum())
    first = rowidx == 0
    end2 = np.where(first, np.minimum(end, np.minimum(ql[jobid], np.maximum(hh[jobid] - 6, 0) + 1)), end)
    inr = (rowidx < kr[jobid]) & dp[jobid]
    base_cost = (kr * cc)[dp].sum()
    for W in (32, 48, 64):
        B = W // 4
        over = (end2 > (beg // B) * B + W) & inr
        fo = np.full(nj, 10 ** 9); np.minimum.at(fo, jobid[over], rowidx[over])
        fits = fo == 10 ** 9
        for T in (6, 8, 10, 12, 16):
            use = dp & (cc > W) & (mm32 >= T)
            cost = np.where(use, np.where(fits, kr * W, np.minimum(fo, kr) * W + kr * cc), kr * cc)
            print(f"  W={W} mm32>={T}: routed {use.sum():6d} fit {(use & fits).sum():6d}  cost ratio {cost[dp].sum() / base_cost:.3f}")

# ---- re-binning at row checkpoints: every K rows a job whose leftmost lane blocks are dead (beg past them) continues in the class that
# is narrower by those blocks; each move is charged MOVE rows of the job's current width.  Upper part of the wedge (end still growing)
# stays as it is.
if len(sys.argv) > 2 and sys.argv[2] == "rebin":
    kr = rules["non-zero frontier + out3 rule"]
    inr = (rowidx < kr[jobid]) & dp[jobid]
    base_cost = (kr * cc)[dp].sum()
    Bw = np.maximum(cc // 4, 8)[jobid]                       # lane block of the job's class (4-lane groups; wider groups: same quarter)
    for K in (16, 32, 64):
        for MOVE in (4, 8, 16):
            ck = (rowidx // K) * K                            # last checkpoint row
            # beg at the last checkpoint: take beg of row ck of the same job
            idx_ck = off[jobid] + np.minimum(ck, rows[jobid] - 1)
            beg_ck = np.where(ck > 0, beg[idx_ck], 0)
            dead = (beg_ck // Bw) * Bw
            width = (cc[jobid] - dead).clip(8)
            cost_rows = (width * inr).sum()
            # number of moves: checkpoints where dead blocks increased
            prev_idx = off[jobid] + np.minimum(np.maximum(ck - K, 0), rows[jobid] - 1)
            dead_prev = np.where(ck - K > 0, (beg[prev_idx] // Bw) * Bw, 0)
            moved = inr & (rowidx == ck) & (ck > 0) & (dead > dead_prev)
            cost_moves = (MOVE * cc[jobid] * moved).sum()
            print(f"  checkpoint every {K:2d} rows, a move costs {MOVE:2d} rows: moves per DP job {moved.sum() / dp.sum():.2f}  cost ratio {(cost_rows + cost_moves) / base_cost:.3f}")
