"""One-off fuzz of the closed-form prefilter: many two-/one-mismatch flanks on random and periodic sequence vs the oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("bwa-mem_gpu_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, d))
import numpy as np
import bwamem_hip as B, oracle_py
from test_gpu_parity import gpu_extend
orc = oracle_py.Oracle()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
tot = bad_tot = 0
for rd in range(rounds):
    rng = np.random.default_rng(1000 + rd)
    qs, ts, h0s = [], [], []
    for it in range(50000):
        ql = int(rng.integers(2, 140)); tl = ql + int(rng.integers(0, ql + 12))
        kind = int(rng.integers(0, 5))
        if kind == 0:
            t = rng.integers(0, 4, size=tl).astype(np.uint8)
        else:
            per = int(rng.integers(1, 9)); unit = rng.integers(0, 4, size=per).astype(np.uint8)
            t = np.tile(unit, tl // per + 1)[:tl]
            if kind == 2:
                k = rng.integers(0, tl, size=max(1, tl // 10)); t[k] = rng.integers(0, 4, size=k.size)
            if kind == 3:
                cut = int(rng.integers(0, tl)); t[:cut] = rng.integers(0, 4, size=cut)
            if kind == 4:
                cut = int(rng.integers(0, tl)); t[cut:] = rng.integers(0, 4, size=tl - cut)
        q = t[:ql].copy()
        nm = int(rng.integers(1, 4))
        pos = rng.integers(0, ql, size=nm)
        if rng.random() < 0.5 and nm >= 2:
            pos[1] = min(ql - 1, pos[0] + int(rng.integers(1, 9)))
        for pp in set(int(x) for x in pos):
            q[pp] = (q[pp] + rng.integers(1, 4)) & 3
        qs.append(q); ts.append(t); h0s.append(int(rng.integers(1, 151)))
    qlen = np.array([len(x) for x in qs], np.uint32); tlen = np.array([len(x) for x in ts], np.uint32)
    qoff = np.concatenate([[0], np.cumsum(qlen)[:-1]]).astype(np.uint32); toff = np.concatenate([[0], np.cumsum(tlen)[:-1]]).astype(np.uint32)
    jobs = (np.concatenate(qs), qoff, qlen, np.concatenate(ts), toff, tlen, np.array(h0s, np.uint32))
    sc = [(1, 4, 6, 1), (2, 3, 5, 2), (1, 3, 4, 2), (3, 2, 6, 1), (1, 2, 3, 1)][rd % 5]
    op = oracle_py.KswParams(sc[0], sc[1], sc[2], sc[3], sc[2], sc[3], 0, 5, 1)
    w3, w6, _ = orc.extend_batch(*jobs, params=op, want_raw=True, n_threads=64)
    g3, g6 = gpu_extend(B, jobs, scoring=sc)
    bad = np.nonzero((g6 != w6).any(1))[0]
    tot += len(qlen); bad_tot += bad.size
    print("round %d scoring %s: %d jobs, %d mismatches" % (rd, sc, len(qlen), bad.size), flush=True)
    if bad.size:
        for i in bad[:3]:
            print(" job", i, "qlen", qlen[i], "h0", h0s[i], "got", g6[i], "want", w6[i])
print("FUZZ TOTAL %d jobs, %d mismatches" % (tot, bad_tot))
