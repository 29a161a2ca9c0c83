"""Differential fuzz of the extension kernels against the oracle's ksw_extend2 on flank-like jobs: a query cut from a "read", a target
that is the same stretch mutated (substitutions, indels, N), optionally with an unrelated tail on either sequence -- the shapes on
which the exact early stop fires --, under several scoring schemes, z-drop settings and end bonuses; raw 6-tuple and the three-result
form.  Test infrastructure (calls oracle/).  usage: fuzz_extend.py [n_jobs_per_round] [rounds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("bwa-mem_gpu_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
import bwamem_hip as B
import oracle_py

n_jobs = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 8
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rng = np.random.default_rng(seed)
oracle_py.build(ref=False)
O = oracle_py.Oracle()


def make_jobs(n, max_q):
    q_parts, t_parts, qlen, tlen, h0 = [], [], [], [], []
    for _ in range(n):
        ql = int(rng.integers(min(MIN_Q, max_q), max_q + 1))
        q = rng.integers(0, 4, size=ql).astype(np.uint8)
        kind = rng.random()
        # the related part of the target: the query's prefix of length k, mutated
        k = ql if kind < 0.5 else int(rng.integers(0, ql + 1))
        t = q[:k].copy()
        rate = rng.choice([0.0, 0.01, 0.03, 0.08, 0.2])
        m = rng.random(k) < rate
        t[m] = (t[m] + rng.integers(1, 4, size=int(m.sum()))) & 3
        if k > 4 and rng.random() < 0.4:                       # an indel or two
            for _i in range(int(rng.integers(1, 3))):
                p = int(rng.integers(1, len(t) - 1)) if len(t) > 2 else 0
                L = int(rng.integers(1, 8))
                t = np.concatenate([t[:p], rng.integers(0, 4, size=L).astype(np.uint8), t[p:]]) if rng.random() < 0.5 else np.concatenate([t[:p], t[p + L:]])
        tail = int(rng.integers(0, max(2, ql // 2 + 40)))       # unrelated bases behind it (the window reaches past the alignment)
        t = np.concatenate([t, rng.integers(0, 4, size=tail).astype(np.uint8)])
        if rng.random() < 0.05 and len(t):
            t[int(rng.integers(0, len(t)))] = 4
        if rng.random() < 0.05:
            q[int(rng.integers(0, ql))] = 4
        if len(t) == 0 and rng.random() < 0.9:
            t = rng.integers(0, 4, size=1).astype(np.uint8)
        q_parts.append(q); t_parts.append(t); qlen.append(ql); tlen.append(len(t)); h0.append(int(rng.integers(1, 150)))
    qlen = np.array(qlen, np.uint32); tlen = np.array(tlen, np.uint32)
    qoff = np.concatenate([[0], np.cumsum(qlen)[:-1]]).astype(np.uint32); toff = np.concatenate([[0], np.cumsum(tlen)[:-1]]).astype(np.uint32)
    return np.concatenate(q_parts), qoff, qlen, (np.concatenate(t_parts) if int(tlen.sum()) else np.zeros(1, np.uint8)), toff, tlen, np.array(h0, np.uint32)


MIN_Q = int(os.environ.get("FUZZ_MINQ", "1"))
SCHEMES = [(1, 4, 6, 1, 6, 1), (2, 5, 4, 2, 7, 1), (1, 1, 1, 1, 1, 1), (3, 9, 11, 3, 5, 2), (1, 4, 6, 1, 6, 1), (1, 2, 3, 1, 2, 2), (4, 4, 10, 1, 10, 1), (1, 4, 0, 1, 0, 1)]
total = 0
t_start = time.time()
for r in range(rounds):
    max_q = [128, 128, 280, 60, 136, 700, 128, 256][r % 8]
    if os.environ.get("FUZZ_MAXQ"): max_q = int(os.environ["FUZZ_MAXQ"])      # (FUZZ_MINQ / FUZZ_MAXQ: every round on one range of query lengths -- one class of kernels)
    jobs = make_jobs(n_jobs if max_q <= 300 else n_jobs // 8, max_q)
    a, b, od, ed, oi, ei = SCHEMES[r % len(SCHEMES)]
    zdrop = [0, 100, 0, 7, 0, 100, 0, 30][r % 8]
    eb = [5, 5, 0, 17, 5, 5, 1, 5][r % 8]
    d = [torch.from_numpy(np.ascontiguousarray(x).astype(np.int64)).to(torch.int32).cuda() if x.dtype == np.uint32 else torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in jobs]
    n = len(jobs[2])
    prm = B.ExtParams(a, b, od, ed, oi, ei, zdrop, eb)
    kp = oracle_py.KswParams(a, b, od, ed, oi, ei, zdrop, eb, 1)
    want3, want6, _ = O.extend_batch(*jobs, params=kp, n_threads=8, want_raw=True)
    for packed in (1, 0):
        was = B.load_library().bmh_extend_set_packed(packed)
        out = torch.zeros(n, 3, dtype=torch.int32, device="cuda"); raw = torch.zeros(n, 6, dtype=torch.int32, device="cuda")
        B.extend_batch(*d, out, params=prm, raw_t=raw)
        out_b = torch.full((n, 3), -77, dtype=torch.int32, device="cuda")
        B.extend_batch(*d, out_b, params=prm, raw_t=None)
        torch.cuda.synchronize()
        B.load_library().bmh_extend_set_packed(was)
        o, o6, ob = out.cpu().numpy(), raw.cpu().numpy(), out_b.cpu().numpy()
        bad = np.flatnonzero((o != want3).any(1) | (o6 != want6).any(1) | (ob != want3).any(1))
        if bad.size:
            i = int(bad[0])
            print("MISMATCH round %d packed %d job %d (of %d bad): qlen %d tlen %d h0 %d scheme %s zdrop %d eb %d\n gpu3 %s raw %s three-result form %s\n want3 %s raw %s" %
                  (r, packed, i, bad.size, jobs[2][i], jobs[5][i], jobs[6][i], SCHEMES[r % 8], zdrop, eb, o[i], o6[i], ob[i], want3[i], want6[i]))
            sys.exit(1)
    total += n
    print("round %d: %d jobs (queries up to %d), scheme %s zdrop %d end bonus %d: identical (packed and 32-bit kernels, raw and three-result forms)  [%.0f s]" %
          (r, n, max_q, SCHEMES[r % 8], zdrop, eb, time.time() - t_start), flush=True)
print("FUZZ OK: %d jobs" % total)
