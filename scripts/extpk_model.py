#!/usr/bin/env python3
"""CPU model of the packed 16-bit extension rows (csrc/extpk_kernels.hip), checked against the oracle's ksw_extend2.

Development tool: it validates the ALGEBRA the packed kernel relies on, column-parallel exactly as the kernel computes it
(numpy vectors over the query columns, unsigned 16-bit saturating arithmetic), before the HIP transcription:
  * no `beg` bookkeeping at all: the first-column value H(i,-1) = max(0, h0 - o_del - e_del*(i+1)) is applied in every row.
    (beg > 0 implies that value has reached zero, and it never comes back; cells left of beg have zero inputs.)
  * M = (hd != 0) * (score + b) + hd  -_sat b      (one mad + one saturating subtract, clamps at zero for free)
  * E, F, M - oe as unsigned saturating values (the reference clamps all three at zero)
  * F by a max-plus prefix scan of g(j) = (M(j) -_sat oe_ins) + e_ins*j
  * only H is masked at and right of `end` (M and E are zero there by themselves; asserted)
  * row maximum + last column on ties from max over (h << 4 | slot) style keys
  * new end = last non-zero H column + 3
usage: extpk_model.py [n_jobs] [seed]
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("oracle", "tests", "bwa-mem_gpu_amd"):
    sys.path.insert(0, os.path.join(ROOT, p))
import oracle_py
import common


def usub(a, b):
    return np.where(a > b, a - b, 0)


def model_extend(q, t, h0, P):
    a, b, od, ed, oi, ei, zdrop = P.a, P.b, P.o_del, P.e_del, P.o_ins, P.e_ins, P.zdrop
    qlen, tlen = len(q), len(t)
    oe_d, oe_i = od + ed, oi + ei
    QP = qlen + 8                                   # a few pad columns on the right, like the kernel's last lane
    j = np.arange(QP)
    qq = np.concatenate([q.astype(np.int64), np.full(QP - qlen, 7)])      # 7 = pad
    H = np.where(j < qlen, np.maximum(h0 - oe_i - j * ei, 0), 0).astype(np.int64)
    E = np.zeros(QP, np.int64)
    end = qlen
    mx, max_i, max_j, max_ie, gscore, max_off = h0, -1, -1, -1, -1, 0
    for i in range(tlen):
        ti = int(t[i])
        hfc = h0 if i == 0 else max(0, h0 - (od + ed * i))
        hnx = max(0, h0 - (od + ed * (i + 1)))
        hd = np.concatenate([[hfc], H[:-1]])
        # score + b as an unsigned byte table: match a+b, mismatch 0, N b-1, pad 0
        scb = np.where(qq > 4, 0, np.where((qq > 3) | (ti > 3), b - 1, np.where(qq == ti, a + b, 0)))
        mask = np.minimum(hd, 1)
        M = usub(mask * scb + hd, b)
        Mi = usub(M, oe_i); Md = usub(M, oe_d)
        g = Mi + ei * j
        cm = np.maximum.accumulate(g)
        F = np.concatenate([[0], np.maximum(cm[:-1] - ei * j[:-1], 0)])          # F(j) = max_{j'<j} g(j') - e*(j-1), clamped
        hraw = np.maximum(np.maximum(M, E), F)
        En = np.maximum(usub(E, ed), Md)
        act = j < end
        assert not (M[(~act) & (j < qlen)]).any(), "M != 0 right of end"
        assert not (E[~act & (j < qlen)]).any(), "E != 0 right of end"
        h = np.where(act, hraw, 0)
        H = h; E = np.where(j < qlen, En, 0)       # (pad columns: the kernel lets E float there; nothing reads it)
        m = int(h.max()) if QP else 0
        if end == qlen:
            h1 = int(h[qlen - 1]) if qlen > 0 else hnx
            if not (gscore > h1): max_ie = i
            gscore = max(gscore, h1)
        if m == 0:
            break
        mj = int(np.nonzero(h == m)[0][-1])
        if m > mx:
            mx, max_i, max_j = m, i, mj
            max_off = max(max_off, abs(mj - i))
        elif zdrop > 0:
            if i - max_i > mj - max_j:
                if mx - m - ((i - max_i) - (mj - max_j)) * ed > zdrop: break
            else:
                if mx - m - ((mj - max_j) - (i - max_i)) * ei > zdrop: break
        last = int(np.nonzero(h)[0][-1])
        end = min(qlen, last + 3)
    return mx, max_j + 1, max_i + 1, max_ie + 1, gscore, max_off


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    orc = oracle_py.Oracle()
    rng = np.random.default_rng(seed)
    scorings = [oracle_py.default_params()]
    for kw in (dict(a=2, b=3, o_del=5, e_del=2, o_ins=4, e_ins=1, zdrop=100), dict(a=1, b=1, o_del=1, e_del=1, o_ins=1, e_ins=1, zdrop=20),
               dict(a=1, b=4, o_del=6, e_del=1, o_ins=6, e_ins=1, zdrop=100), dict(a=3, b=2, o_del=2, e_del=3, o_ins=7, e_ins=2, zdrop=0)):
        p = oracle_py.default_params()
        for k, v in kw.items(): setattr(p, k, v)
        scorings.append(p)
    bad = 0
    for si, P in enumerate(scorings):
        jobs = common.make_ext_jobs(n, rng, maxq=160)
        q, qoff, qlen, t, toff, tlen, h0 = jobs
        _, want6, _ = orc.extend_batch(*jobs, params=P, want_raw=True)
        for k in range(n):
            got = model_extend(q[qoff[k]:qoff[k] + qlen[k]], t[toff[k]:toff[k] + tlen[k]], int(h0[k]), P)
            if tuple(int(x) for x in want6[k]) != got:
                bad += 1
                if bad < 10: print("MISMATCH scoring", si, "job", k, "qlen", qlen[k], "tlen", tlen[k], "h0", h0[k], "want", want6[k], "got", got)
        print(f"scoring {si}: {n} jobs checked, mismatches so far {bad}", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
