"""CPU tests of the oracle (our C restatement): against the committed golden vectors (produced by the
reference's own compiled C, tests/golden/make_golden.py) and, where oracle/_ref/libref.so exists,
against the reference itself on fresh seeded inputs."""
import os

import numpy as np
import pytest

import common
import oracle_py
from bwamem_hip import fmindex, synth

G = common.GOLDEN


def _golden_genome():
    z = np.load(os.path.join(G, "seed_kat.npz"))
    n = int(z["n_genome"])
    bits = np.unpackbits(z["genome"])[: 2 * n].reshape(n, 2)
    return (bits[:, 0] * 2 + bits[:, 1]).astype(np.uint8), z


def test_ext_golden(oracle):
    z = np.load(os.path.join(G, "ext_kat.npz"))
    jobs = tuple(z[k] for k in ("q", "qoff", "qlen", "t", "toff", "tlen", "h0"))
    for zd in (0, 100):
        o3, r6, cells = oracle.extend_batch(*jobs, params=oracle_py.default_params(zdrop=zd), want_raw=True)
        assert np.array_equal(r6, z[f"raw6_z{zd}"])
        assert np.array_equal(o3, z[f"out3_z{zd}"])
        assert cells > 0


def test_seed_golden(oracle):
    g, z = _golden_genome()
    assert np.array_equal(g, synth.make_genome(200_000, seed=42)), "generator drifted from the golden genome"
    idx = fmindex.build_fmd_index(g)
    got = oracle.seed_reads(oracle.fmd(idx), z["reads"], z["offs"], z["lens"], 19)
    for k in ("smem_k", "smem_s", "smem_qb", "smem_qe", "smem_read") + common.SEED_KEYS:
        assert np.array_equal(got[k], z[k]), k
    # threads do not change the result
    got4 = oracle.seed_reads(oracle.fmd(idx), z["reads"], z["offs"], z["lens"], 19, n_threads=4)
    common.assert_seeds_equal(got4, got)


def test_occ_sa_golden(oracle):
    g, _ = _golden_genome()
    idx = fmindex.build_fmd_index(g)
    f = oracle.fmd(idx)
    z = np.load(os.path.join(G, "occ_kat.npz"))
    for k, occ, sa in zip(z["k"], z["occ"], z["sa"]):
        for c in range(4):
            assert oracle.lib.fmd_occ(f, int(k), c) == int(occ[c])
        assert oracle.lib.fmd_sa(f, int(k), None) == int(sa)


def test_seed_properties(oracle):
    """Size-independent properties: every located seed is an exact match of the read in the
    fwd+revcomp text; SMEMs of a read are strictly increasing in begin and end; prefix = scan."""
    g, idx = common.genome_and_index(150_000, seed=9)
    text = np.concatenate([g, synth.revcomp(g)])
    reads, _ = synth.make_reads(g, 600, 150, seed=5)
    flat, offs, lens = common.flat_reads(reads)
    s = oracle.seed_reads(oracle.fmd(idx), flat, offs, lens)
    assert np.array_equal(np.cumsum(s["n_ref_pos"])[:-1], s["prefix"][1:])
    for r in range(len(lens)):
        lo, n = int(s["prefix"][r]), int(s["n_ref_pos"][r])
        i, last = lo, (-1, -1)
        while i < lo + n:
            cnt = int(s["score"][i]); b, e = s["qbeg"][i]
            assert cnt >= 1 and e - b >= 19 and b > last[0] and e > last[1]
            last = (b, e)
            for t in range(cnt):
                p = int(s["rbeg"][i + t])
                assert np.array_equal(text[p:p + e - b], reads[r, b:e])
                assert tuple(s["qbeg"][i + t]) == (b, e) and (t == 0 or s["score"][i + t] == 0)
            i += cnt


def test_oracle_vs_ref_fresh(oracle, ref):
    g, idx = common.genome_and_index(120_000, seed=77)
    b = ref.bwt_from_index(idx)
    reads, _ = synth.make_reads(g, 1200, 150, seed=78, sub_rate=0.03)
    rows = [r for r in reads] + common.edge_reads(g, np.random.default_rng(8))
    flat, offs, lens = common.ragged_reads(rows)
    for k in (19, 10):
        common.assert_seeds_equal(oracle.seed_reads(oracle.fmd(idx), flat, offs, lens, k), ref.seed_reads(b, flat, offs, lens, k))
    # the re-interleaved form of the same index (what bench.py's cpu_baseline hands the reference's code at hg38 scale)
    for n_g in (120_000, 120_037):
        g2, idx2 = common.genome_and_index(n_g, seed=77)
        b2 = ref.bwt_from_index_fast(idx2)
        r2, _ = synth.make_reads(g2, 300, 150, seed=5)
        f2, o2, l2 = common.flat_reads(r2)
        common.assert_seeds_equal(oracle.seed_reads(oracle.fmd(idx2), f2, o2, l2, 19), ref.seed_reads(b2, f2, o2, l2, 19))
    jobs = common.make_ext_jobs(1500, np.random.default_rng(79))
    for zd in (0, 30):
        p = oracle_py.default_params(zdrop=zd)
        o3, r6, _ = oracle.extend_batch(*jobs, params=p, want_raw=True)
        ro3, rr6 = ref.extend_batch(*jobs, params=p)
        assert np.array_equal(r6, rr6) and np.array_equal(o3, ro3)


def test_ext_edge_cases(oracle, ref):
    """empty query, single cells, all-N, all-mismatch, h0 = 1, very long target."""
    rows = [
        (np.zeros(0, np.uint8), np.array([0, 1, 2], np.uint8), 10),
        (np.array([0], np.uint8), np.array([0], np.uint8), 1),
        (np.array([0], np.uint8), np.array([1], np.uint8), 1),
        (np.full(50, 4, np.uint8), np.full(60, 4, np.uint8), 30),
        (np.zeros(40, np.uint8), np.full(90, 3, np.uint8), 25),
        (np.tile(np.arange(4, dtype=np.uint8), 30), np.tile(np.arange(4, dtype=np.uint8), 100), 19),
        (np.tile(np.arange(4, dtype=np.uint8), 70), np.tile(np.arange(4, dtype=np.uint8), 10), 150),
    ]
    q, qoff, qlen = common.ragged_reads([r[0] for r in rows])
    t, toff, tlen = common.ragged_reads([r[1] for r in rows])
    if q.size == 0:
        q = np.zeros(1, np.uint8)
    h0 = np.array([r[2] for r in rows], np.uint32)
    jobs = (q, qoff.astype(np.uint32), qlen, t, toff.astype(np.uint32), tlen, h0)
    o3, r6, _ = oracle.extend_batch(*jobs, want_raw=True)
    ro3, rr6 = ref.extend_batch(*jobs)
    assert np.array_equal(r6, rr6) and np.array_equal(o3, ro3)


def test_global_alignment_restatement_matches_reference(oracle, ref):
    """oracle_ksw_global2 (score and CIGAR, i.e. every direction bit the traceback visits) == the reference's compiled
    ksw_global2 (src/ksw.c:1120) on seeded cases with substitutions, indels, N bases, two scorings, bands >= |tlen - qlen|
    (the only bands bwa_gen_cigar2 produces: w >= |rlen - l_query| + 3, src/bwa.c:160-166)."""
    import oracle_py
    rng = np.random.default_rng(3)
    for it in range(4000):
        ql = int(rng.integers(1, 160))
        t = rng.integers(0, 4, size=ql + int(rng.integers(0, 30))).astype(np.uint8)
        q = t[:ql].copy()
        for _ in range(int(rng.integers(0, 4))):
            pos = int(rng.integers(0, len(q))); q[pos] = (q[pos] + rng.integers(1, 4)) & 3
        if it % 3 == 0 and len(q) > 10:
            pos = int(rng.integers(1, len(q) - 1)); k = int(rng.integers(1, 6))
            q = np.concatenate([q[:pos], rng.integers(0, 4, size=k).astype(np.uint8), q[pos:]]) if it % 2 else np.concatenate([q[:pos], q[pos + k:]])
        if it % 40 == 0 and len(q) > 0:
            q[int(rng.integers(0, len(q)))] = 4
        if it % 7 == 0:
            t = t[: max(1, len(q) + int(rng.integers(-5, 6)))]
        if len(q) == 0:
            continue
        d = abs(len(t) - len(q))
        w = d + int(rng.integers(0, 60)) if it % 5 else d + 3
        p = oracle_py.default_params() if it % 4 else oracle_py.KswParams(2, 3, 5, 2, 4, 1, 0, 5, 1)
        a = oracle.global2(q, t, w, p); b = ref.global2(q, t, w, p)
        assert a[0] == b[0] and np.array_equal(a[1], b[1]), (it, len(q), len(t), w)


def test_reg2aln_restatement_matches_reference_sam(oracle):
    """oracle_reg2aln (mem_reg2aln + bwa_gen_cigar2 restated) reproduces POS, CIGAR, NM, MD and strand of every primary
    SAM record the reference's own host code wrote for the golden read set (scripts/make_jobs_golden.py)."""
    from bwamem_hip import synth
    from bwamem_hip.lib import HostJobs
    z = np.load(os.path.join(common.GOLDEN, "jobs_golden.npz"))
    g = synth.make_genome(int(z["n_genome"]), seed=int(z["genome_seed"]))
    reads = z["reads"]; n, L = reads.shape
    seeds = {k: z[k] for k in ("rbeg", "qbeg", "score", "n_ref_pos", "prefix")}
    hj = HostJobs(g, reads.reshape(-1), np.arange(n, dtype=np.uint64) * L, np.full(n, L, np.uint32), seeds, n_threads=4)
    out3, _, _ = oracle.extend_batch(*hj.jobs())
    regs = hj.merge(out3)
    pad = (-len(g)) % 4
    codes = np.concatenate([g, np.zeros(pad, np.uint8)]).reshape(-1, 4)
    pac = np.ascontiguousarray(((codes[:, 0] << 6) | (codes[:, 1] << 4) | (codes[:, 2] << 2) | codes[:, 3]).astype(np.uint8))
    checked = n_indel = 0
    for r in range(n):
        rr = regs[regs[:, 0] == r]
        if len(rr) == 0 or z["as_tag"][r] < 0:
            continue
        cand = np.unique(rr[rr[:, 1] == rr[:, 1].max()][:, 1:], axis=0)      # the primary = the best region
        if len(cand) != 1:
            continue
        c = cand[0]
        rb = int(np.uint32(c[3])) | (int(c[4]) << 32); re = int(np.uint32(c[5])) | (int(c[6]) << 32)
        a = oracle.reg2aln(pac, len(g), reads[r], c[1], c[2], rb, re, c[0])
        cs = "".join(f"{int(x) >> 4}{'MIDSH'[int(x) & 0xf]}" for x in a["cigar"])
        got = (a["pos"] + 1, cs, a["NM"], a["MD"], a["is_rev"])
        want = (int(z["sam_pos"][r]), str(z["sam_cigar"][r]), int(z["sam_nm"][r]), str(z["sam_md"][r]), (int(z["sam_flag"][r]) >> 4) & 1)
        assert got == want, (r, got, want)
        checked += 1; n_indel += ("I" in cs) or ("D" in cs)
    hj.free()
    assert checked > 0.95 * n and n_indel > 50
