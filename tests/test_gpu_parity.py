"""GPU parity tests proper: HIP kernels (through the C ABI) vs the CPU oracle, bit-exact."""
import os
import numpy as np
import pytest

import common

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import torch
    import bwamem_hip as B
    B.load_library()           # raises if the HIP extension is missing: no fallback
    assert torch.cuda.is_available(), "these tests need a GPU"
    return B


def _to_dev(torch, a, dt=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dt is not None:
        t = t.view(dt) if t.dtype.itemsize == torch.empty(0, dtype=dt).element_size() else t.to(dt)
    return t.cuda()


def gpu_seed(B, idx, flat, offs, lens, min_seed_len=19, densify=None, genome=None, max_occ=1 << 22, index=None):
    import torch
    from bwamem_hip.lib import seeds_to_host
    from bwamem_hip import synth
    ascii_ = synth.codes_to_ascii(flat) if flat.size else np.zeros(1, np.uint8)
    # genome given: the 2-bit text goes up too, which switches on the unique-interval shortcuts of the seeding kernels
    dindex = index if index is not None else (B.Index.upload(idx) if genome is None else B.Index.upload(idx, pac=_pack_pac(genome), l_pac=len(genome)))
    if densify:
        dindex.densify_sa(densify)
    ws = B.SeedWorkspace(max(len(lens), 1), max(int(flat.size), 1), max_cands=max(int(flat.size), 64), max_occ=max_occ)
    r = _to_dev(torch, ascii_)
    o = torch.from_numpy(offs.astype(np.int64)).to(torch.int32).cuda()
    l = torch.from_numpy(lens.astype(np.int64)).to(torch.int32).cuda()
    s = ws.seed_batch(dindex, r, o, l, min_seed_len)
    out = seeds_to_host(s, len(lens))
    out["n_smems"] = int(s.n_smems)
    ws.free()
    if index is None:
        dindex.free()
    return out


@pytest.fixture(params=["split", "fused"])
def pipeline(request):
    """Both seeding pipelines of libbwamem_hip.so (the choice is read once per process, so the fused
    one runs in a child process)."""
    return request.param


def _run_fused_child(test_name):
    import os, subprocess, sys
    env = dict(os.environ, BMH_SEED_FUSED="1", BMH_CHILD="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.abspath(__file__) + "::" + test_name + "[split]"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()[-3000:]


def test_seeding_matches_oracle_150bp(hip, oracle, pipeline):
    import os
    if pipeline == "fused":
        if os.environ.get("BMH_CHILD"):
            pytest.skip("already inside the fused child")
        return _run_fused_child("test_seeding_matches_oracle_150bp")
    g, idx = common.genome_and_index(300_000)
    reads, _ = hip.synth.make_reads(g, 4000, 150, seed=11)
    flat, offs, lens = common.flat_reads(reads)
    want = oracle.seed_reads(oracle.fmd(idx), flat, offs, lens)
    got = gpu_seed(hip, idx, flat, offs, lens)
    assert got["n_smems"] == len(want["smem_k"])
    common.assert_seeds_equal(got, want)


def test_seeding_edge_cases(hip, oracle, pipeline):
    import os
    if pipeline == "fused":
        if os.environ.get("BMH_CHILD"):
            pytest.skip("already inside the fused child")
        return _run_fused_child("test_seeding_edge_cases")
    g, idx = common.genome_and_index(300_000)
    rows = common.edge_reads(g, np.random.default_rng(3))
    flat, offs, lens = common.ragged_reads(rows)
    want = oracle.seed_reads(oracle.fmd(idx), flat, offs, lens)
    got = gpu_seed(hip, idx, flat, offs, lens)
    common.assert_seeds_equal(got, want)


def test_seeding_300bp_and_other_k(hip, oracle):
    g, idx = common.genome_and_index(300_000)
    reads, _ = hip.synth.make_reads(g, 1500, 300, seed=12)
    flat, offs, lens = common.flat_reads(reads)
    for k in (19, 25, 12):
        want = oracle.seed_reads(oracle.fmd(idx), flat, offs, lens, min_seed_len=k)
        got = gpu_seed(hip, idx, flat, offs, lens, min_seed_len=k)
        common.assert_seeds_equal(got, want, what=f"k={k} ")


def test_seeding_odd_genome_length(hip, oracle):
    g, idx = common.genome_and_index(100_003, seed=5)
    reads, _ = hip.synth.make_reads(g, 1000, 101, seed=13)
    flat, offs, lens = common.flat_reads(reads)
    want = oracle.seed_reads(oracle.fmd(idx), flat, offs, lens)
    got = gpu_seed(hip, idx, flat, offs, lens)
    common.assert_seeds_equal(got, want)


def test_seeding_grows_its_occurrence_arrays(hip, oracle):
    """More located seeds than the workspace was sized for (reads from a 300-copy repeat, capacity 64): the output arrays are
    replaced by larger ones instead of failing."""
    rng = np.random.default_rng(3)
    g = rng.integers(0, 4, size=400_000).astype(np.uint8)
    rep = rng.integers(0, 4, size=400).astype(np.uint8)
    for p in rng.integers(0, len(g) - 400, size=300):
        g[p:p + 400] = rep
    from bwamem_hip import fmindex
    idx = fmindex.build_fmd_index(g)
    reads = np.stack([rep[s:s + 101] for s in rng.integers(0, 299, size=200)])
    flat, offs, lens = common.flat_reads(reads)
    want = oracle.seed_reads(oracle.fmd(idx), flat, offs, lens)
    assert len(want["rbeg"]) > 20000
    got = gpu_seed(hip, idx, flat, offs, lens, max_occ=64)
    common.assert_seeds_equal(got, want)


@pytest.mark.parametrize("new_intv", [4, 1])
def test_seeding_with_denser_sa_samples(hip, oracle, new_intv):
    """bmh_index_densify_sa: the samples of every 4th row / of every row computed on the device from the file's (every 16th)
    -- located positions (and everything else) unchanged, odd text length, 33rd-bit values included."""
    g, idx = common.genome_and_index(100_003, seed=5)
    reads, _ = hip.synth.make_reads(g, 1500, 101, seed=14)
    flat, offs, lens = common.flat_reads(reads)
    want = oracle.seed_reads(oracle.fmd(idx), flat, offs, lens)
    got = gpu_seed(hip, idx, flat, offs, lens, densify=new_intv)
    common.assert_seeds_equal(got, want)


@pytest.mark.parametrize("case", ["150bp", "repeats_N", "short_genome", "sparse_sa"])
def test_seeding_unique_interval_shortcut(hip, oracle, case):
    """With the 2-bit text resident the seeding kernels stop ranking once an interval holds a single suffix and compare the
    read with the text instead (16 symbols per load): same SMEMs, same positions -- reads with N, reads from repeats,
    matches that run into the end of either strand or across the strand boundary (tiny genome), read ends, sparse SA."""
    rng = np.random.default_rng(77)
    if case == "short_genome":
        g, idx = common.genome_and_index(2_003, seed=9)
        reads, _ = hip.synth.make_reads(g, 3000, 101, seed=15, sub_rate=0.005)
    else:
        g, idx = common.genome_and_index(300_001, seed=8)
        reads, _ = hip.synth.make_reads(g, 4000, 150, seed=16, sub_rate=0.02 if case == "repeats_N" else 0.01)
    if case == "repeats_N":
        reads[::7, 40] = 4; reads[::11, 0] = 4; reads[::13, -1] = 4; reads[::17, 60:63] = 4
        reads[5::50] = rng.integers(0, 4, size=reads[5::50].shape)              # reads from nowhere
        rep = g[1000:1150].copy(); reads[3::40] = rep                            # many identical reads
    flat, offs, lens = common.flat_reads(reads)
    want = oracle.seed_reads(oracle.fmd(idx), flat, offs, lens)
    got = gpu_seed(hip, idx, flat, offs, lens, genome=g, densify=None if case == "sparse_sa" else 1)
    common.assert_seeds_equal(got, want)
    # the backward search in phases (the walks that are still searching after k steps parked and resumed in full waves: an
    # experiment knob, off by default): same seeds, also with a phase per step and with lists that meet out of order
    for phases in ("1", "2,3,5", "6,12,24,48"):
        os.environ["BMH_SEED_BWD_PHASES"] = phases
        try:
            got = gpu_seed(hip, idx, flat, offs, lens, genome=g, densify=None if case == "sparse_sa" else 1)
        finally:
            os.environ.pop("BMH_SEED_BWD_PHASES", None)
        common.assert_seeds_equal(got, want, "phases " + phases + ": ")


def gpu_extend(B, jobs, zdrop=0, want_raw=True, scoring=None, packed=None):
    """packed: None = the library's default routing (packed 16-bit kernels for the jobs that qualify), 0 = 32-bit kernels only"""
    import torch
    if packed is not None:
        was = B.load_library().bmh_extend_set_packed(int(packed))
        try:
            return gpu_extend(B, jobs, zdrop, want_raw, scoring)
        finally:
            B.load_library().bmh_extend_set_packed(was)
    q, qoff, qlen, t, toff, tlen, h0 = jobs
    n = len(qlen)
    d = [torch.from_numpy(np.ascontiguousarray(x).astype(np.int64) if x.dtype == np.uint32 else np.ascontiguousarray(x)) for x in (q, qoff, qlen, t, toff, tlen, h0)]
    d = [x.to(torch.int32).cuda() if x.dtype == torch.int64 else x.cuda() for x in d]
    out = torch.zeros(n, 3, dtype=torch.int32, device="cuda")
    raw = torch.zeros(n, 6, dtype=torch.int32, device="cuda") if want_raw else None
    prm = B.ExtParams.default(zdrop=zdrop)
    if scoring is not None:
        if len(scoring) == 6:
            a_, b_, od_, ed_, oi_, ei_ = scoring
            prm = B.ExtParams(a_, b_, od_, ed_, oi_, ei_, zdrop, 5)
        else:
            a_, b_, o_, e_ = scoring
            prm = B.ExtParams(a_, b_, o_, e_, o_, e_, zdrop, 5)
    B.extend_batch(*d, out, params=prm, raw_t=raw)
    torch.cuda.synchronize()
    # both launch forms of the packed kernels -- one kernel per class, and the persistent kernel that works all classes off
    # (csrc/extpk_dev.h: extpk_persist_kernel; knob EXT_PERSIST = blocks per CU) -- must give the same numbers
    L_ = B.load_library()
    for persist in (0, 1, 3):
        L_.bmh_tune_set(b"EXT_PERSIST", persist, 0)
        try:
            out_p = torch.zeros(n, 3, dtype=torch.int32, device="cuda")         # (zeros like `out`: an unsupported job's raw tuple is left alone)
            raw_p = torch.zeros(n, 6, dtype=torch.int32, device="cuda") if want_raw else None
            B.extend_batch(*d, out_p, params=prm, raw_t=raw_p)
            torch.cuda.synchronize()
        finally:
            L_.bmh_tune_set(b"EXT_PERSIST", 0, 1)
        assert torch.equal(out_p, out) and (not want_raw or torch.equal(raw_p, raw)), f"EXT_PERSIST={persist} differs from the default launch form"
    if want_raw:
        # the production form (three results per job, no raw 6-tuple) may stop a job earlier -- as soon as the local-vs-to-end
        # rule is decided -- and must return the same three numbers
        out_b = torch.full((n, 3), -77, dtype=torch.int32, device="cuda")
        B.extend_batch(*d, out_b, params=prm, raw_t=None)
        torch.cuda.synchronize()
        diff = (out_b != out).any(1).nonzero().flatten()[:5].cpu().numpy()
        assert diff.size == 0, f"three-result form differs from the raw form at {diff}: {out_b.cpu().numpy()[diff]} vs {out.cpu().numpy()[diff]}"
    return out.cpu().numpy(), raw.cpu().numpy() if want_raw else None


@pytest.mark.parametrize("packed", [1, 0])
@pytest.mark.parametrize("zdrop", [0, 100])
def test_extension_matches_oracle(hip, oracle, zdrop, packed):
    import oracle_py
    jobs = common.make_ext_jobs(6000, np.random.default_rng(21))
    want3, want6, _ = oracle.extend_batch(*jobs, params=oracle_py.default_params(zdrop=zdrop), want_raw=True)
    got3, got6 = gpu_extend(hip, jobs, zdrop=zdrop, packed=packed)
    bad = np.nonzero((got6 != want6).any(1))[0]
    assert bad.size == 0, f"{bad.size} raw mismatches, first {bad[:5]}: got {got6[bad[:5]]} want {want6[bad[:5]]} qlen {jobs[2][bad[:5]]} tlen {jobs[5][bad[:5]]}"
    assert np.array_equal(got3, want3)


@pytest.mark.parametrize("scoring", [(2, 5, 7, 2), (3, 31, 9, 3), (33, 40, 50, 7)])
def test_extension_other_scorings(hip, oracle, scoring):
    """Scores from the bit-field table (|a|,|b| < 32) and from the generic compare/select form (a = 33), with N bases in
    queries and targets (make_ext_jobs plants them), z-drop on."""
    import oracle_py
    a, b, o, e = scoring
    jobs = common.make_ext_jobs(3000, np.random.default_rng(31 + a))
    h0 = (jobs[6].astype(np.int64) * a).astype(np.uint32)          # seed scores scale with the match score
    jobs = jobs[:6] + (h0,)
    want3, want6, _ = oracle.extend_batch(*jobs, params=oracle_py.KswParams(a, b, o, e, o, e, 100 * a, 5, 1), want_raw=True)
    got3, got6 = gpu_extend(hip, jobs, zdrop=100 * a, scoring=scoring)
    bad = np.nonzero((got6 != want6).any(1))[0]
    assert bad.size == 0, f"{bad.size} raw mismatches, first {bad[:5]}: got {got6[bad[:5]]} want {want6[bad[:5]]} qlen {jobs[2][bad[:5]]} tlen {jobs[5][bad[:5]]}"
    assert np.array_equal(got3, want3)


@pytest.mark.parametrize("packed", [1, 0])
@pytest.mark.parametrize("scoring", [(1, 4, 6, 1, 4, 2), (2, 3, 5, 2, 9, 1), (1, 1, 1, 1, 2, 1)])
def test_extension_asymmetric_gap_penalties(hip, oracle, scoring, packed):
    """Different open / extend penalties for deletions and insertions (the packed kernels compute M - oe twice then), N bases,
    z-drop on and off, on both kernel families."""
    import oracle_py
    a, b, od, ed, oi, ei = scoring
    for zdrop in (0, 60):
        jobs = common.make_ext_jobs(2500, np.random.default_rng(77 + a + zdrop), maxq=288)
        want3, want6, _ = oracle.extend_batch(*jobs, params=oracle_py.KswParams(a, b, od, ed, oi, ei, zdrop, 5, 1), want_raw=True)
        got3, got6 = gpu_extend(hip, jobs, zdrop=zdrop, scoring=scoring, packed=packed)
        bad = np.nonzero((got6 != want6).any(1))[0]
        assert bad.size == 0, f"{bad.size} raw mismatches, first {bad[:5]}: got {got6[bad[:5]]} want {want6[bad[:5]]} qlen {jobs[2][bad[:5]]} tlen {jobs[5][bad[:5]]}"
        assert np.array_equal(got3, want3)


def test_extension_packed_class_boundaries(hip, oracle):
    """Jobs at the edges of what the packed 16-bit kernels take (csrc/extpk_dev.h): query lengths around every group / pair-count
    boundary (128|129 columns: 16 -> 17 pairs on 4 lanes, 136|137: 4 -> 8 lanes, 160|161: 8 lanes of 10 pairs -> 4 lanes of 24, 256|257: 4 lanes of 32 pairs -> 8 of 18,
    288|289: packed -> 32-bit), targets around the LDS staging caps (384, 512, 640) and seed scores that push h0 + qlen*a across the 4096 limit of the 16-bit keys (2048 where a
    lane holds more than 16 pairs: 5-bit pair indices) -- the router must send each job to a kernel that is exact for it."""
    import oracle_py
    rng = np.random.default_rng(99)
    qs, ts, h0 = [], [], []
    def add(ql, tl, h):
        t = rng.integers(0, 4, size=tl).astype(np.uint8)
        q = np.resize(t, ql).copy() if ql else np.zeros(0, np.uint8)
        if ql:
            mut = rng.random(ql) < 0.06
            q[mut] = (q[mut] + rng.integers(1, 4, size=int(mut.sum()))) & 3
        qs.append(q); ts.append(t); h0.append(h)
    for ql in (1, 15, 16, 17, 31, 32, 33, 64, 65, 96, 97, 112, 113, 127, 128, 129, 132, 135, 136, 137, 143, 144, 145, 159, 160, 161, 192, 193, 224, 225, 255, 256, 257, 287, 288, 289, 300):
        for tl in (ql + 7, 383, 384, 385, 511, 512, 513, 639, 640, 641):
            for h in (1, 19, 150, 2048 - ql - 1, 2048 - ql, 2048 - ql + 1, 4096 - ql - 1, 4096 - ql, 4096 - ql + 1, 5000):       # (2048: the 17-pair class of 129..136 columns)
                add(ql, tl, h)
    qlen = np.array([len(x) for x in qs], np.uint32); tlen = np.array([len(x) for x in ts], np.uint32)
    qoff = np.concatenate([[0], np.cumsum(qlen)[:-1]]).astype(np.uint32); toff = np.concatenate([[0], np.cumsum(tlen)[:-1]]).astype(np.uint32)
    jobs = (np.concatenate(qs), qoff, qlen, np.concatenate(ts), toff, tlen, np.array(h0, np.uint32))
    for zdrop in (0, 100):
        want3, want6, _ = oracle.extend_batch(*jobs, params=oracle_py.default_params(zdrop=zdrop), want_raw=True)
        got3, got6 = gpu_extend(hip, jobs, zdrop=zdrop)
        bad = np.nonzero((got6 != want6).any(1))[0]
        assert bad.size == 0, f"{bad.size} raw mismatches, first {bad[:5]}: got {got6[bad[:5]]} want {want6[bad[:5]]} qlen {jobs[2][bad[:5]]} tlen {jobs[5][bad[:5]]} h0 {jobs[6][bad[:5]]}"
        assert np.array_equal(got3, want3)


def test_extension_jobs_without_target_rows(hip, oracle):
    """tlen == 0 (the window of a seed at the end of a sequence is clipped away, src/bntseq.c:531-556): the answer is
    (h0, 0, 0).  Such jobs sort last in their class; a class made only of them, one of them alone in the batch, and
    mixtures with ordinary jobs in every kernel form (static, job-drawing, wide)."""
    rng = np.random.default_rng(41)
    def batch(qlens, tlens):
        qs = [rng.integers(0, 4, size=q).astype(np.uint8) for q in qlens]
        ts = [np.concatenate([qs[i][:min(len(qs[i]), t)], rng.integers(0, 4, size=max(t - len(qs[i]), 0)).astype(np.uint8)])[:t] for i, t in enumerate(tlens)]
        qlen = np.array(qlens, np.uint32); tlen = np.array(tlens, np.uint32)
        qoff = np.concatenate([[0], np.cumsum(qlen)[:-1]]).astype(np.uint32); toff = np.concatenate([[0], np.cumsum(tlen)[:-1]]).astype(np.uint32)
        cat = lambda xs: np.concatenate(xs) if sum(len(x) for x in xs) else np.zeros(1, np.uint8)
        return (cat(qs), qoff, qlen, cat(ts), toff, tlen, rng.integers(19, 100, size=len(qlens)).astype(np.uint32))
    cases = [([70], [0]),                                                # one job, no rows (was lost: the loop ended before its result was written)
             ([3, 70, 85], [4, 0, 5]),
             ([5, 20, 40, 60, 70, 100, 130, 200, 280, 300, 400], [0] * 11),            # every class, only empty targets
             (list(range(1, 300, 7)) * 3, [0 if i % 3 == 0 else int(rng.integers(1, 400)) for i in range(len(range(1, 300, 7)) * 3)])]
    for qlens, tlens in cases:
        jobs = batch(qlens, tlens)
        want3, want6, _ = oracle.extend_batch(*jobs, want_raw=True)
        got3, got6 = gpu_extend(hip, jobs)
        assert np.array_equal(got3, want3), (qlens[:12], tlens[:12], got3[:12], want3[:12])
        assert np.array_equal(got6, want6)


def test_extension_long_queries(hip, oracle):
    jobs = common.make_ext_jobs(600, np.random.default_rng(22), maxq=512)
    want3, want6, _ = oracle.extend_batch(*jobs, want_raw=True)
    got3, got6 = gpu_extend(hip, jobs)
    assert np.array_equal(got6, want6) and np.array_equal(got3, want3)
    # up to the supported maximum of 768 bases (flanks of the reads the chaining stages admit)
    jobs = common.make_ext_jobs(500, np.random.default_rng(24), maxq=768)
    assert (jobs[2] > 512).sum() > 50
    want3, want6, _ = oracle.extend_batch(*jobs, want_raw=True)
    got3, got6 = gpu_extend(hip, jobs)
    assert np.array_equal(got6, want6) and np.array_equal(got3, want3)
    assert hip.load_library().bmh_extend_last_unsupported() == 0
    # beyond it: marked, counted, never silently wrong
    rng = np.random.default_rng(25)
    q = rng.integers(0, 4, size=769 + 30, dtype=np.uint8); t = rng.integers(0, 4, size=900 + 40, dtype=np.uint8)
    jobs = (q, np.array([0, 769], np.uint32), np.array([769, 30], np.uint32), t, np.array([0, 900], np.uint32), np.array([900, 40], np.uint32), np.array([30, 25], np.uint32))
    got3, _ = gpu_extend(hip, jobs)
    assert (got3[0] == np.iinfo(np.int32).min).all() and hip.load_library().bmh_extend_last_unsupported() == 1
    want3, _, _ = oracle.extend_batch(*jobs)
    assert np.array_equal(got3[1], want3[1])
    # short queries against very long targets (beyond the LDS staging of the 16-lane-row kernel)
    rng = np.random.default_rng(23)
    qs, ts, h0 = [], [], []
    for it in range(200):
        tl = int(rng.integers(1000, 1600)); ql = int(rng.integers(20, 280))
        t = rng.integers(0, 4, size=tl).astype(np.uint8)
        q = t[:ql].copy()
        for _ in range(int(rng.integers(0, 6))):
            p = int(rng.integers(0, ql)); q[p] = (q[p] + rng.integers(1, 4)) & 3
        if it % 3 == 0:
            k = int(rng.integers(1, 8)); p = int(rng.integers(5, ql - 5)); q = np.concatenate([q[:p], q[p + k:]])
        qs.append(q); ts.append(t); h0.append(int(rng.integers(19, 120)))
    qlen = np.array([len(x) for x in qs], np.uint32); tlen = np.array([len(x) for x in ts], np.uint32)
    qoff = np.concatenate([[0], np.cumsum(qlen)[:-1]]).astype(np.uint32); toff = np.concatenate([[0], np.cumsum(tlen)[:-1]]).astype(np.uint32)
    jobs = (np.concatenate(qs), qoff, qlen, np.concatenate(ts), toff, tlen, np.array(h0, np.uint32))
    want3, want6, _ = oracle.extend_batch(*jobs, want_raw=True)
    got3, got6 = gpu_extend(hip, jobs)
    assert np.array_equal(got6, want6) and np.array_equal(got3, want3)


def test_seed_gpu_file_api(hip, oracle, tmp_path):
    """The reference's own call sequence (fastmap.c:436-465) through include/seed_gen.h."""
    from bwamem_hip import fmindex, synth
    g, idx = common.genome_and_index(300_000)
    prefix = str(tmp_path / "ref.fa")
    fmindex.write_index(prefix, idx)
    reads, _ = synth.make_reads(g, 2500, 150, seed=14)
    fq = str(tmp_path / "reads.fa")
    synth.write_fasta_reads(fq, reads)
    got = hip.seed_file(prefix, fq, 19)
    flat, offs, lens = common.flat_reads(reads)
    want = oracle.seed_reads(oracle.fmd(idx), flat, offs, lens)
    common.assert_seeds_equal(got, want)


def test_gasal_shim_matches_oracle(hip, oracle, tmp_path):
    """The GASAL2-source-compatible API (include/gasal2_root) driven like the reference's host code."""
    import os, struct, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "gasal_shim_driver")
    subprocess.check_call(["g++", "-O1", "-std=c++11", "-fpermissive", "-w", "-I", os.path.join(root, "include", "gasal2_root", "src"),
                           os.path.join(root, "tests", "gasal_shim_driver.cpp"), "-o", exe,
                           "-L", os.path.join(root, "bwa-mem_gpu_amd"), "-lbwamem_hip", "-Wl,-rpath," + os.path.join(root, "bwa-mem_gpu_amd"),
                           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"])
    jobs = common.make_ext_jobs(3001, np.random.default_rng(31), maxq=150)
    q, qoff, qlen, t, toff, tlen, h0 = jobs
    inp, outp = str(tmp_path / "jobs.bin"), str(tmp_path / "res.bin")
    with open(inp, "wb") as f:
        f.write(struct.pack("<I", len(qlen)))
        for a in (qoff, qlen, toff, tlen, h0):
            f.write(np.ascontiguousarray(a, dtype="<u4").tobytes())
        f.write(struct.pack("<I", q.size)); f.write(q.tobytes())
        f.write(struct.pack("<I", t.size)); f.write(t.tobytes())
    subprocess.check_call([exe, inp, outp])
    got = np.fromfile(outp, dtype="<i4").reshape(-1, 3)
    want3, _, _ = oracle.extend_batch(*jobs)
    assert np.array_equal(got, want3)


def test_reference_gase_aln_end_to_end(hip, tmp_path):
    """BASELINE configs[0] shape through the REFERENCE's own host code: build/dropin/bwa-gasal2 is the
    reference's src/*.c compiled unchanged against include/seed_gen.h + include/gasal2_root and linked
    with libbwamem_hip.so (scripts/build_dropin.sh, build container only)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "build", "dropin", "bwa-gasal2")):
        pytest.skip("build/dropin/bwa-gasal2 not built (needs /root/reference at build time)")
    # single-end (configs[0] shape), interleaved paired-end with -p (configs[3] shape), and hard pairs (diverged / relocated /
    # random / chimeric mates) under a non-default option set given to both sides
    # (the first run is configs[0] at its stated size: 10 000 reads vs a 4.64 Mbp genome, -t 1)
    for size, extra in ((["4640000", "10000"], []), (["2000000", "4000"], ["1", "pe"]),
                        (["2000000", "4000"], ["1", "pe_hard", "-k 21 -B 6 -O 8,9 -E 2,3 -T 50 -U 25 -m 20 -M -Y -a"])):
        r = subprocess.run([sys.executable, os.path.join(root, "scripts", "e2e_dropin.py"), str(tmp_path)] + size + extra,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        out = r.stdout.decode()
        assert r.returncode == 0 and "E2E DROP-IN OK" in out, out[-3000:]
        assert "SAM IDENTICAL" in out, out[-3000:]           # bwamem_hip.aligner wrote the reference's records byte for byte
    # the reference's real mode: -t 8, every kt_for worker with its own 2 x 2 gasal_gpu_storage_t (src/kthread.c:158-161).  On the clean
    # read set its output is deterministic and equals ours ...
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "e2e_dropin.py"), str(tmp_path), "2000000", "4000", "8"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    out = r.stdout.decode()
    assert r.returncode == 0 and "SAM IDENTICAL" in out, out[-3000:]
    # ... and on a hard set (4 % substitutions, indels, chimeric reads, every 7th read across a 12-60 bp deletion) it does so once the four
    # batch-relative seq[] indices of mem_align1_core (src/bwamem.c:2228, 2251, 2295, 2313) are made absolute -- the build
    # bwa-gasal2-seqidx of scripts/build_dropin.sh; as shipped, :2313 hands mem_sort_dedup_patch ANOTHER read's bases, which a second
    # worker may be converting in place at that moment (:2052-2053): the garbage scores of INTEGRATION.md section 2
    if os.path.exists(os.path.join(root, "build", "dropin", "bwa-gasal2-seqidx")):
        env = dict(os.environ, E2E_EXE="bwa-gasal2-seqidx", E2E_LONGDEL="7", E2E_TAG="seqidx_t8")
        r = subprocess.run([sys.executable, os.path.join(root, "scripts", "e2e_dropin.py"), str(tmp_path), "2000000", "30000", "8", "se_hard"],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env)
        out = r.stdout.decode()
        assert r.returncode == 0 and "SAM IDENTICAL" in out, out[-3000:]


def test_dropin_with_several_device_workers(hip, tmp_path):
    """BMH_DEVICES=2 (SURVEY.md 8e from C: one host worker thread per device; on a one-GPU box both workers share it): seed_gpu
    deals the batches of the read file to two workers -- the second one on its own replica of the index (bmh_index_replicate when
    its device differs) -- and the gasal storages are spread over the devices.  The reference binary must write the same SAM."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "build", "dropin", "bwa-gasal2")):
        pytest.skip("build/dropin/bwa-gasal2 not built (needs /root/reference at build time)")
    env = dict(os.environ, BMH_DEVICES="2", BMH_SEED_BATCH_READS="700")
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "e2e_dropin.py"), str(tmp_path), "2000000", "4000", "2"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env)
    out = r.stdout.decode()
    assert r.returncode == 0 and "E2E DROP-IN OK" in out and "SAM IDENTICAL" in out, out[-3000:]


def test_index_replicate_and_shard_range(hip, oracle):
    """bmh_index_replicate: a second copy of the index (same device here; device to device over xGMI on a node) seeds identically;
    bmh_shard_range equals bwamem_hip.parallel.shard_range."""
    import ctypes as C
    import torch
    from bwamem_hip.parallel import shard_range
    L = hip.load_library()
    L.bmh_shard_range.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.bmh_shard_range.restype = None
    for n, world, mult in ((10, 3, 1), (1000001, 8, 2), (7, 8, 1), (0, 2, 2), (999, 4, 2)):
        for rank in range(world):
            lo, hi = C.c_uint64(), C.c_uint64()
            L.bmh_shard_range(n, rank, world, mult, C.byref(lo), C.byref(hi))
            assert (lo.value, hi.value) == shard_range(n, rank, world, mult)
    g, idx = common.genome_and_index(100_000)
    reads, _ = hip.synth.make_reads(g, 600, 150, seed=3)
    flat, offs, lens = common.flat_reads(reads)
    want = oracle.seed_reads(oracle.fmd(idx), flat, offs, lens)
    index = hip.Index.upload(idx, pac=_pack_pac(g), l_pac=len(g))
    L.bmh_index_replicate.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    rep = C.c_void_p()
    assert L.bmh_index_replicate(index.handle, 0, 0, C.byref(rep)) == 0 and rep.value
    copy = hip.Index(rep.value)
    got = gpu_seed(hip, idx, flat, offs, lens, index=copy)
    common.assert_seeds_equal(got, want, "replicated index: ")
    copy.free(); index.free()


def test_host_job_builder_matches_reference_host_code(hip, tmp_path):
    """bmh_build_jobs (chain -> chain_flt -> chain2aln restatement) vs the job stream the REFERENCE's own host
    code submits (build/dropin/bwa-gasal2 with BMH_GASAL_DUMP), and best region score vs the SAM AS tag."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "build", "dropin", "bwa-gasal2")):
        pytest.skip("build/dropin/bwa-gasal2 not built (needs /root/reference at build time)")
    # (the -W 5 and the 800 bp runs go through the reference's seed filter mem_flt_chained_seeds / mem_seed_sw, src/bwamem.c:970-991)
    for args in (["1500000", "3000", "150"], ["1500000", "1500", "300"], ["1500000", "2000", "150", "-W", "5"], ["1500000", "300", "740"]):
        r = subprocess.run([sys.executable, os.path.join(root, "scripts", "check_jobs_vs_reference.py"), str(tmp_path)] + args,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        out = r.stdout.decode()
        assert r.returncode == 0 and "JOBS VS REFERENCE OK" in out, out[-3000:]


@pytest.mark.parametrize("zdrop", [0, 100, 3])
def test_extension_closed_form_jobs(hip, oracle, zdrop):
    """Jobs the closed-form prefilter decides without DP (<= 1 substitution, no gap, no N) and their near misses:
    mismatch at every kind of position, tiny h0 (the diagonal dies at the mismatch), tlen == qlen, N bases,
    two mismatches, low-complexity sequence where off-diagonal paths are strong."""
    import oracle_py
    rng = np.random.default_rng(77)
    qs, ts, h0s = [], [], []
    for it in range(6000):
        ql = int(rng.integers(1, 132))
        tl = ql + int(rng.integers(0, ql + 20))
        kind = it % 8
        if kind == 7:
            t = np.full(tl, rng.integers(0, 4), np.uint8)            # homopolymer
        elif kind == 6:
            t = np.tile(rng.integers(0, 4, size=2).astype(np.uint8), tl // 2 + 1)[:tl]   # dinucleotide repeat
        else:
            t = rng.integers(0, 4, size=tl).astype(np.uint8)
        q = t[:ql].copy()
        nm = [0, 1, 1, 1, 2, 1, 1, 1][kind]
        if nm >= 1:
            pos = [0, ql - 1, int(rng.integers(0, ql))][it % 3]
            q[pos] = (q[pos] + rng.integers(1, 4)) & 3
        if nm == 2:
            pos2 = int(rng.integers(0, ql)); q[pos2] = (q[pos2] + rng.integers(1, 4)) & 3
        if it % 50 == 0:
            q[int(rng.integers(0, ql))] = 4
        if it % 50 == 1:
            t[int(rng.integers(0, tl))] = 4
        qs.append(q); ts.append(t)
        h0s.append(int(rng.integers(1, 8)) if it % 5 == 0 else int(rng.integers(1, 151)))
    qlen = np.array([len(x) for x in qs], np.uint32); tlen = np.array([len(x) for x in ts], np.uint32)
    qoff = np.concatenate([[0], np.cumsum(qlen)[:-1]]).astype(np.uint32); toff = np.concatenate([[0], np.cumsum(tlen)[:-1]]).astype(np.uint32)
    jobs = (np.concatenate(qs), qoff, qlen, np.concatenate(ts), toff, tlen, np.array(h0s, np.uint32))
    want3, want6, _ = oracle.extend_batch(*jobs, params=oracle_py.default_params(zdrop=zdrop), want_raw=True)
    got3, got6 = gpu_extend(hip, jobs, zdrop=zdrop)
    bad = np.nonzero((got6 != want6).any(1))[0]
    assert bad.size == 0, f"{bad.size} mismatches, first {bad[:5]}: got {got6[bad[:5]]} want {want6[bad[:5]]} qlen {qlen[bad[:5]]} tlen {tlen[bad[:5]]} h0 {np.array(h0s)[bad[:5]]}"
    assert np.array_equal(got3, want3)


def test_extension_two_mismatch_closed_form(hip, oracle):
    """Two-substitution flanks (closed form + its shifted-diagonal guard): random sequence, tandem repeats of
    period 1..6 (where a gapped path can tie or beat the diagonal), mismatches at every spacing, short h0."""
    rng = np.random.default_rng(99)
    qs, ts, h0s = [], [], []
    for it in range(40000):
        ql = int(rng.integers(2, 132))
        tl = ql + int(rng.integers(0, ql + 12))
        kind = it % 4
        if kind == 0:
            t = rng.integers(0, 4, size=tl).astype(np.uint8)
        else:
            per = int(rng.integers(1, 7))
            unit = rng.integers(0, 4, size=per).astype(np.uint8)
            t = np.tile(unit, tl // per + 1)[:tl]
            if kind == 2:                                   # repeat with a few random bases sprinkled in
                k = rng.integers(0, tl, size=max(1, tl // 12)); t[k] = rng.integers(0, 4, size=k.size)
            if kind == 3:                                   # random left part, periodic right part
                cut = int(rng.integers(0, tl)); t[:cut] = rng.integers(0, 4, size=cut)
        q = t[:ql].copy()
        p1 = int(rng.integers(0, ql))
        gap = int(rng.integers(1, 12)) if it % 3 else int(rng.integers(1, ql + 1))
        p2 = min(ql - 1, p1 + gap)
        for pp in {p1, p2}:
            q[pp] = (q[pp] + rng.integers(1, 4)) & 3
        qs.append(q); ts.append(t)
        h0s.append(int(rng.integers(1, 12)) if it % 7 == 0 else int(rng.integers(1, 151)))
    qlen = np.array([len(x) for x in qs], np.uint32); tlen = np.array([len(x) for x in ts], np.uint32)
    qoff = np.concatenate([[0], np.cumsum(qlen)[:-1]]).astype(np.uint32); toff = np.concatenate([[0], np.cumsum(tlen)[:-1]]).astype(np.uint32)
    jobs = (np.concatenate(qs), qoff, qlen, np.concatenate(ts), toff, tlen, np.array(h0s, np.uint32))
    import oracle_py
    # default scoring with and without z-drop, then other scorings (different Lmax / Lg of the closed-form guard)
    for (a, b, o, e, zdrop) in ((1, 4, 6, 1, 0), (1, 4, 6, 1, 7), (2, 3, 5, 2, 0), (1, 3, 4, 2, 0), (3, 2, 6, 1, 0), (1, 1, 6, 1, 0), (1, 4, 6, 1, 3)):
        op = oracle_py.KswParams(a, b, o, e, o, e, zdrop, 5, 1)
        want3, want6, _ = oracle.extend_batch(*jobs, params=op, want_raw=True, n_threads=8)
        got3, got6 = gpu_extend(hip, jobs, zdrop=zdrop, scoring=(a, b, o, e))
        bad = np.nonzero((got6 != want6).any(1))[0]
        assert bad.size == 0, f"scoring {(a, b, o, e, zdrop)}: {bad.size} mismatches, first {bad[:5]}: got {got6[bad[:5]]} want {want6[bad[:5]]} qlen {qlen[bad[:5]]} h0 {np.array(h0s)[bad[:5]]}"
        assert np.array_equal(got3, want3)


def _pack_pac(g):
    pad = (-len(g)) % 4
    codes = np.concatenate([g, np.zeros(pad, np.uint8)]).reshape(-1, 4)
    pac = ((codes[:, 0] << 6) | (codes[:, 1] << 4) | (codes[:, 2] << 2) | codes[:, 3]).astype(np.uint8)
    return np.ascontiguousarray(np.concatenate([pac, np.zeros(1, np.uint8)]))      # + the .pac tail byte slot


def _device_chain_case(B, oracle, g, idx, reads, opt_over=None, heavy=None, scoring=False, sub=None):
    """reads -> bmh_seed_batch -> bmh_chain_batch -> bmh_extend_batch -> bmh_chain_merge, all in HBM, against the
    host job builder on the same seeds (byte-identical batch) and its merge of the oracle's extension results."""
    import ctypes as C, os, torch
    from bwamem_hip import synth
    from bwamem_hip.lib import ChainOpt, ChainWorkspace, HostJobs, dev_jobs_to_host, seeds_to_host, load_library
    if isinstance(reads, np.ndarray):
        n, L = reads.shape
        flat = np.ascontiguousarray(reads.reshape(-1))
        offs_h = np.arange(n, dtype=np.uint64) * L; lens_h = np.full(n, L, np.uint32)
    else:                                                    # ragged: a list of reads of different lengths
        flat, offs_h, lens_h = common.ragged_reads(reads)
        n = len(reads)
    dindex = B.Index.upload(idx, pac=_pack_pac(g), l_pac=len(g))
    ws = B.SeedWorkspace(n, max(int(flat.size), 1), max_cands=max(int(flat.size), 64), max_occ=1 << 22)
    r = _to_dev(torch, synth.codes_to_ascii(flat))
    o = torch.from_numpy(offs_h.astype(np.int64)).to(torch.int32).cuda()
    l = torch.from_numpy(lens_h.astype(np.int64)).to(torch.int32).cuda()
    s = ws.seed_batch(dindex, r, o, l, 19)
    opt = ChainOpt(); load_library().bmh_chain_opt_default(C.byref(opt))
    for k, v in (opt_over or {}).items():
        setattr(opt, k, v)
    cw = ChainWorkspace(n, max(int(s.n_seeds), 1), opt=opt)
    # scoring: the extension runs with the chain options' scores too (otherwise with the defaults, whatever opt_over says)
    ext_p = B.ExtParams(opt.a, opt.b, opt.o_del, opt.e_del, opt.o_ins, opt.e_ins, 0, 5) if scoring else B.ExtParams.default()
    import oracle_py
    ksw_p = oracle_py.KswParams(ext_p.a, ext_p.b, ext_p.o_del, ext_p.e_del, ext_p.o_ins, ext_p.e_ins, 0, 5, 1)
    # heavy: reads with more sampled seeds than this leave the lane kernel; sub: mask of the classes (<= 16 / 32 / 64 entries) chained four reads per wave
    # (knob CHAIN_SUB; 0 = their round-5 forms: a lane per read over the class list, a wave per read)
    if heavy is not None:
        os.environ["BMH_CHAIN_HEAVY"] = str(heavy)
    if sub is not None:
        os.environ["BMH_CHAIN_SUB"] = str(sub)
    try:
        dj = cw.chain_batch(dindex, r, o, l, s)
    finally:
        os.environ.pop("BMH_CHAIN_HEAVY", None)
    got = dev_jobs_to_host(dj, n)
    hj = HostJobs(g, flat, offs_h, lens_h, seeds_to_host(s, n), n_threads=4, opt=opt)
    assert int(dj.n_jobs) == hj.n_jobs and int(dj.n_regs) == hj.n_regs
    for k in ("qlen", "tlen", "h0", "job_read", "job_reg", "job_side", "qoff", "toff", "regs_per_read", "q", "t"):
        assert np.array_equal(got[k], getattr(hj, k)), k
    assert np.array_equal(got["frac_rep"], hj.frac_rep())
    # extension + merge on the device vs oracle extension + host merge
    out3 = torch.zeros(max(hj.n_jobs, 1), 3, dtype=torch.int32, device="cuda")
    regs = torch.zeros(max(hj.n_regs, 1), 8, dtype=torch.int32, device="cuda")
    if hj.n_jobs:
        rc = load_library().bmh_extend_batch(dj.d_q, dj.d_qoff, dj.d_qlen, dj.d_t, dj.d_toff, dj.d_tlen, dj.d_h0, int(dj.n_jobs),
                                             C.byref(ext_p), out3.data_ptr(), None, None)
        assert rc == 0
    cw.merge(out3, regs)
    torch.cuda.synchronize()
    want3, _, _ = oracle.extend_batch(*hj.jobs(), params=ksw_p) if hj.n_jobs else (np.zeros((0, 3), np.int32), None, None)
    assert np.array_equal(out3.cpu().numpy()[: hj.n_jobs], want3)
    assert np.array_equal(regs.cpu().numpy()[: hj.n_regs], hj.merge(want3))
    # the same without materialised base arrays: bmh_chain_extend reads the bases from the reads / 2-bit reference
    cw.set_materialize(False)
    dj2 = cw.chain_batch(dindex, r, o, l, s)
    assert int(dj2.n_jobs) == hj.n_jobs and not dj2.d_q and not dj2.d_t
    out3b = torch.full((max(hj.n_jobs, 1), 3), -7, dtype=torch.int32, device="cuda")
    raw6 = torch.zeros(max(hj.n_jobs, 1), 6, dtype=torch.int32, device="cuda")
    regs2 = torch.zeros(max(hj.n_regs, 1), 8, dtype=torch.int32, device="cuda")
    cw.extend(out3b, params=ext_p, raw_t=raw6)
    cw.merge(out3b, regs2)
    torch.cuda.synchronize()
    assert np.array_equal(out3b.cpu().numpy()[: hj.n_jobs], want3)
    assert np.array_equal(regs2.cpu().numpy()[: hj.n_regs], hj.merge(want3))
    # the one-call form (two passes, the heavy reads' chaining hidden behind the first pass's extension): same regions, read order
    regs3 = torch.full((hj.n_regs + 3, 8), -9, dtype=torch.int32, device="cuda")
    dj3 = cw.extend_merge(dindex, r, o, l, s, regs3, params=ext_p)
    torch.cuda.synchronize()
    assert int(dj3.n_jobs) == hj.n_jobs and int(dj3.n_regs) == hj.n_regs
    assert np.array_equal(regs3.cpu().numpy()[: hj.n_regs], hj.merge(want3)) and (regs3.cpu().numpy()[hj.n_regs:] == -9).all()
    tm = cw.extend_merge_timing()
    assert tm["jobs_a"] + tm["jobs_b"] == hj.n_jobs
    if hj.n_regs > 1:
        with pytest.raises(RuntimeError, match="capacity"):
            cw.extend_merge(dindex, r, o, l, s, regs3[: hj.n_regs - 1])
    stats = (hj.n_jobs, hj.n_regs, int(dj.n_heavy_reads))
    os.environ.pop("BMH_CHAIN_SUB", None)
    hj.free(); cw.free(); ws.free(); dindex.free()
    return stats


def test_device_job_builder_matches_host_builder(hip, oracle):
    """bmh_chain_batch (chains, chain filter, chain2aln and the on-device reference fetch) produces the same batch,
    byte for byte, as bmh_build_jobs -- itself pinned to the reference's host code -- in the lane form, in the wave
    form (every read forced through it), for 300 bp reads, non-default options and a repeat-rich genome."""
    from bwamem_hip import fmindex, synth
    g, idx = common.genome_and_index(1_500_000)
    reads, _ = synth.make_reads(g, 3000, 150, seed=5, sub_rate=0.02, indel_frac=0.2)
    nj, nr, nh = _device_chain_case(hip, oracle, g, idx, reads)
    assert nj > 3000 and nr > 2500
    _, _, nh = _device_chain_case(hip, oracle, g, idx, reads[:1500], heavy=0)          # all reads by the cooperative forms: four reads per wave up to 64 entries
    assert nh > 1400
    _, _, nh = _device_chain_case(hip, oracle, g, idx, reads[:1500], heavy=0, sub=0)   # ... a lane per read over the class lists / a wave per read (the round-5 forms)
    assert nh > 1400
    _device_chain_case(hip, oracle, g, idx, reads[:1500], heavy=0, sub=5)              # ... mixed
    reads3, _ = synth.make_reads(g, 1000, 300, seed=6, sub_rate=0.03, indel_frac=0.3)
    _device_chain_case(hip, oracle, g, idx, reads3)
    _device_chain_case(hip, oracle, g, idx, reads[:1500], opt_over=dict(max_occ=3, max_chain_extend=2, min_chain_weight=25, drop_ratio=0.8))
    # the reference's seed filter (mem_flt_chained_seeds / mem_seed_sw, src/bwamem.c:970-991, 774-807) on the device: 1.1 W <= 0.05 l
    # (W = 5 and 6 at 150 bp, 13 at 300 bp), lane and wave forms, default and other scores (the local alignment's and the extension's)
    _device_chain_case(hip, oracle, g, idx, reads[:1500], opt_over=dict(min_chain_weight=5))
    _device_chain_case(hip, oracle, g, idx, reads[:800], opt_over=dict(min_chain_weight=5), heavy=0)
    _device_chain_case(hip, oracle, g, idx, reads[:800], opt_over=dict(min_chain_weight=5), heavy=0, sub=0)
    _device_chain_case(hip, oracle, g, idx, reads3[:500], opt_over=dict(min_chain_weight=13))
    _device_chain_case(hip, oracle, g, idx, reads[:1000], opt_over=dict(min_chain_weight=6, a=2, b=5, o_del=4, e_del=2, o_ins=7, e_ins=1), scoring=True)
    # repeat-rich genome: many copies, little divergence -> reads with hundreds of seeds and chains
    gr = synth.make_genome(600_000, seed=9, repeat_frac=0.6, repeat_len=(200, 800), repeat_copies=(50, 400), repeat_div=0.02)
    idxr = fmindex.build_fmd_index(gr)
    readsr, _ = synth.make_reads(gr, 2000, 150, seed=8, sub_rate=0.01)
    nj, nr, nh = _device_chain_case(hip, oracle, gr, idxr, readsr)
    assert nh > 20, nh
    _device_chain_case(hip, oracle, gr, idxr, readsr, sub=0)
    _device_chain_case(hip, oracle, gr, idxr, readsr, heavy=0)                          # the single-seed reads of a repeat-rich genome through the four-per-wave form too
    _device_chain_case(hip, oracle, gr, idxr, readsr[:800], opt_over=dict(min_chain_weight=5))      # seed-rich reads through the filter
    # ragged batch: reads of 30..250 bases (some shorter than a seed), N bases, the edge cases of the seeding tests
    rng = np.random.default_rng(31)
    rows = common.edge_reads(g, rng)
    for _ in range(600):
        ln = int(rng.integers(30, 251)); p = int(rng.integers(0, len(g) - ln))
        x = g[p:p + ln].copy()
        m = rng.random(ln) < 0.03; x[m] = (x[m] + rng.integers(1, 4, size=int(m.sum()))) & 3
        if rng.random() < 0.5:
            x = synth.revcomp(x)
        if rng.random() < 0.1:
            x[int(rng.integers(0, ln))] = 4
        rows.append(x)
    _device_chain_case(hip, oracle, g, idx, rows)
    _device_chain_case(hip, oracle, gr, idxr, readsr[:600], heavy=4, opt_over=dict(max_occ=20))
    # reads from the first and the last bases of the reference, both strands: their target windows are clipped at position 0, at
    # l_pac and at 2 l_pac, where the packed kernels' eight-rows-per-lane fetch of the 2-bit text clamps its three-byte window
    rows = []
    for k in range(240):
        ln = int(rng.integers(60, 200)); p = int(rng.integers(0, 12)) if k & 1 else len(g) - ln - int(rng.integers(0, 12))
        x = g[p:p + ln].copy()
        m = rng.random(ln) < 0.04; x[m] = (x[m] + rng.integers(1, 4, size=int(m.sum()))) & 3
        rows.append(synth.revcomp(x) if k & 2 else x)
    _device_chain_case(hip, oracle, g, idx, rows)


def test_device_job_builder_matches_reference_job_stream(hip):
    """Direct pin of the DEVICE job builder: the seeds recorded with the reference's own run (tests/golden/jobs_golden.npz,
    scripts/make_jobs_golden.py) go straight into bmh_chain_batch; the extension jobs it builds are, as a multiset of
    (h0, query, target), the jobs the reference's host code submitted to its extension library, and after bmh_extend_batch +
    bmh_chain_merge the best region score of every read equals the AS tag of the reference's SAM output."""
    import ctypes as C, hashlib, torch
    from bwamem_hip import fmindex, synth
    from bwamem_hip.lib import ChainWorkspace, SeedsT, dev_jobs_to_host, load_library
    B = hip
    z = np.load(os.path.join(common.GOLDEN, "jobs_golden.npz"))
    g = synth.make_genome(int(z["n_genome"]), seed=int(z["genome_seed"]))
    reads = z["reads"]
    n, L = reads.shape
    dindex = B.Index.upload(fmindex.build_fmd_index(g), pac=_pack_pac(g), l_pac=len(g))
    dev = {k: torch.from_numpy(np.ascontiguousarray(z[k]).view(np.int64 if z[k].dtype == np.uint64 else np.int32)).cuda() for k in ("rbeg", "qbeg", "score", "n_ref_pos", "prefix")}
    s = SeedsT()
    s.n_seeds = len(z["rbeg"]); s.n_smems = int((z["score"] > 0).sum()); s.n_cands = 0
    s.d_rbeg, s.d_qbeg, s.d_score, s.d_n_ref_pos, s.d_prefix = (dev[k].data_ptr() for k in ("rbeg", "qbeg", "score", "n_ref_pos", "prefix"))
    r = _to_dev(torch, synth.codes_to_ascii(reads.reshape(-1)))
    o = (torch.arange(n, dtype=torch.int64) * L).to(torch.int32).cuda()
    l = torch.full((n,), L, dtype=torch.int32).cuda()
    cw = ChainWorkspace(n, max(int(s.n_seeds), 1))
    dj = cw.chain_batch(dindex, r, o, l, s)
    got = dev_jobs_to_host(dj, n)
    nj = int(dj.n_jobs)
    digs = sorted(hashlib.sha1(bytes([int(got["h0"][i]) & 255, int(got["h0"][i]) >> 8]) + got["q"][got["qoff"][i]:got["qoff"][i] + got["qlen"][i]].tobytes() + b"|" +
                               got["t"][got["toff"][i]:got["toff"][i] + got["tlen"][i]].tobytes()).digest() for i in range(nj))
    assert digs == [bytes(x) for x in z["job_digests"]]
    out3 = torch.zeros(nj + 1, 3, dtype=torch.int32, device="cuda")
    regs = torch.zeros(int(dj.n_regs) + 1, 8, dtype=torch.int32, device="cuda")
    assert load_library().bmh_extend_batch(dj.d_q, dj.d_qoff, dj.d_qlen, dj.d_t, dj.d_toff, dj.d_tlen, dj.d_h0, nj, C.byref(B.ExtParams.default()), out3.data_ptr(), None, None) == 0
    cw.merge(out3, regs)
    torch.cuda.synchronize()
    rg = regs.cpu().numpy()[: int(dj.n_regs)]
    best = np.full(n, -1, np.int64)
    np.maximum.at(best, rg[:, 0], rg[:, 1])
    has = z["as_tag"] >= 0
    assert has.sum() > 0.9 * n and np.array_equal(best[has], z["as_tag"][has])
    cw.free(); dindex.free()


def test_device_job_builder_with_alt_contigs_matches_reference_job_stream(hip):
    """The same pin on a genome with ALT contigs (tests/golden/alt_golden.npz: two primary sequences and three ALT contigs named in
    the .alt file): mem_chain_flt does not let a kept ALT chain shadow a chain on the primary assembly (src/bwamem.c:518), so the
    job stream depends on the flags -- with them (bmh_chain_set_alt) it is the reference's, without them it is not."""
    import ast, hashlib, torch
    from bwamem_hip import fmindex, synth
    from bwamem_hip.aligner import read_alt
    from bwamem_hip.lib import ChainWorkspace, SeedsT, dev_jobs_to_host
    B = hip
    z = np.load(os.path.join(common.GOLDEN, "alt_golden.npz"))
    n_g = int(z["n_genome"]); bits = np.unpackbits(z["genome_packed"])[: 2 * n_g].reshape(n_g, 2)
    g = (bits[:, 0] * 2 + bits[:, 1]).astype(np.uint8)
    contigs = ast.literal_eval(str(z["contigs"]))
    names = [c[0] for c in contigs]
    alt = np.array([1 if nm in ("altA1", "altB1", "altA2") else 0 for nm in names], np.uint8)
    reads = z["reads"]
    n, L = reads.shape
    dindex = B.Index.upload(fmindex.build_fmd_index(g), pac=_pack_pac(g), l_pac=len(g))
    dev = {k: torch.from_numpy(np.ascontiguousarray(z[k]).view(np.int64 if z[k].dtype == np.uint64 else np.int32)).cuda() for k in ("rbeg", "qbeg", "score", "n_ref_pos", "prefix")}
    s = SeedsT()
    s.n_seeds = len(z["rbeg"]); s.n_smems = int((z["score"] > 0).sum()); s.n_cands = 0
    s.d_rbeg, s.d_qbeg, s.d_score, s.d_n_ref_pos, s.d_prefix = (dev[k].data_ptr() for k in ("rbeg", "qbeg", "score", "n_ref_pos", "prefix"))
    r = _to_dev(torch, synth.codes_to_ascii(reads.reshape(-1)))
    o = (torch.arange(n, dtype=torch.int64) * L).to(torch.int32).cuda()
    l = torch.full((n,), L, dtype=torch.int32).cuda()

    def digests(with_alt):
        cw = ChainWorkspace(n, max(int(s.n_seeds), 1))
        cw.set_contigs(contigs)
        if with_alt:
            cw.set_alt(alt)
        dj = cw.chain_batch(dindex, r, o, l, s)
        got = dev_jobs_to_host(dj, n)
        d = sorted(hashlib.sha1(bytes([int(got["h0"][i]) & 255, int(got["h0"][i]) >> 8]) + got["q"][got["qoff"][i]:got["qoff"][i] + got["qlen"][i]].tobytes() + b"|" +
                                got["t"][got["toff"][i]:got["toff"][i] + got["tlen"][i]].tobytes()).digest() for i in range(int(dj.n_jobs)))
        cw.free()
        return d
    want = [bytes(x) for x in z["job_digests"]]
    assert digests(True) == want
    assert digests(False) != want, "the golden read set does not exercise the ALT rule of the chain filter"
    # the host builder under the same flags (bmh_chain_opt_t.contig_is_alt)
    from bwamem_hip.lib import ChainOpt, HostJobs, load_library
    import ctypes as C
    copt = ChainOpt(); load_library().bmh_chain_opt_default(C.byref(copt)); copt.contig_is_alt = alt.ctypes.data
    seeds_h = {k: z[k] for k in ("rbeg", "qbeg", "score", "n_ref_pos", "prefix")}
    hj = HostJobs(g, reads.reshape(-1), np.arange(n, dtype=np.uint64) * L, np.full(n, L, np.uint32), seeds_h, n_threads=2, contigs=contigs, opt=copt)
    q, qoff, qlen, t, toff, tlen, h0 = hj.jobs()
    d_host = sorted(hashlib.sha1(bytes([int(h0[i]) & 255, int(h0[i]) >> 8]) + q[qoff[i]:qoff[i] + qlen[i]].tobytes() + b"|" + t[toff[i]:toff[i] + tlen[i]].tobytes()).digest() for i in range(hj.n_jobs))
    hj.free()
    assert d_host == want
    dindex.free()


def _cigar_case(B, oracle, g, idx, reads, scoring=None, max_regs=4000):
    """bmh_cigar_batch on the regions of the device pipeline (every region, not only the best ones) vs the oracle's
    mem_reg2aln restatement (pinned to the reference's SAM output and to its compiled ksw_global2)."""
    import ctypes as C, torch
    import oracle_py
    from bwamem_hip import synth
    from bwamem_hip.lib import ChainWorkspace, cigar_batch
    n, L = reads.shape
    flat = np.ascontiguousarray(reads.reshape(-1))
    pac = _pack_pac(g)
    dindex = B.Index.upload(idx, pac=pac, l_pac=len(g))
    ws = B.SeedWorkspace(n, n * L, max_cands=n * L, max_occ=1 << 22)
    r = _to_dev(torch, synth.codes_to_ascii(flat))
    o = (torch.arange(n, dtype=torch.int64) * L).to(torch.int32).cuda()
    l = torch.full((n,), L, dtype=torch.int32).cuda()
    s = ws.seed_batch(dindex, r, o, l, 19)
    cw = ChainWorkspace(n, max(int(s.n_seeds), 1))
    cw.set_materialize(False)
    dj = cw.chain_batch(dindex, r, o, l, s)
    nr = int(dj.n_regs)
    out3 = torch.zeros(max(int(dj.n_jobs), 1), 3, dtype=torch.int32, device="cuda")
    regs = torch.zeros(max(nr, 1), 8, dtype=torch.int32, device="cuda")
    ep = B.ExtParams.default() if scoring is None else B.ExtParams(*scoring)
    cw.extend(out3, params=ep)
    cw.merge(out3, regs)
    nr = min(nr, max_regs)
    cigar, aln, md = cigar_batch(dindex, r, o, l, regs, nr, params=ep, max_cigar=48, md_cap=640)
    torch.cuda.synchronize()
    cigar = cigar.cpu().numpy().view(np.uint32); aln = aln.cpu().numpy(); md = md.cpu().numpy(); rg = regs.cpu().numpy()
    kp = oracle_py.default_params() if scoring is None else oracle_py.KswParams(*scoring, 1)
    n_gap = 0
    for i in range(nr):
        c = rg[i]
        rb = int(np.uint32(c[4])) | (int(c[5]) << 32); re = int(np.uint32(c[6])) | (int(c[7]) << 32)
        want = oracle.reg2aln(pac, len(g), reads[c[0]], c[2], c[3], rb, re, c[1], params=kp)
        a = aln[i]
        pos = int(np.uint32(a[0])) | (int(a[1]) << 32)
        got_md = bytes(md[i][: a[6]]).decode()
        assert a[7] == 0, (i, a)
        assert (pos, int(a[2]), int(a[4]), int(a[5])) == (want["pos"], want["is_rev"], want["NM"], want["score"]), (i, a, want, c)
        assert np.array_equal(cigar[i][: a[3]], want["cigar"]), (i, cigar[i][: a[3]], want["cigar"], c)
        assert got_md == want["MD"], (i, got_md, want["MD"])
        n_gap += int(((want["cigar"] & 0xf) == 1).any() or ((want["cigar"] & 0xf) == 2).any())
    cw.free(); ws.free(); dindex.free()
    return nr, n_gap


def test_cigar_batch_matches_oracle(hip, oracle):
    """Region -> CIGAR / NM / MD (bmh_cigar_batch) against the oracle's restatement of mem_reg2aln / bwa_gen_cigar2 /
    ksw_global2: 150 bp with indels, 300 bp, a non-default scoring, both strands, clipped and full-length regions."""
    from bwamem_hip import synth
    g, idx = common.genome_and_index(1_500_000)
    reads, _ = synth.make_reads(g, 1500, 150, seed=11, sub_rate=0.02, indel_frac=0.5)
    nr, ngap = _cigar_case(hip, oracle, g, idx, reads)
    assert nr > 1500 and ngap > 300
    reads3, _ = synth.make_reads(g, 400, 300, seed=12, sub_rate=0.03, indel_frac=0.6)
    _cigar_case(hip, oracle, g, idx, reads3)
    _cigar_case(hip, oracle, g, idx, reads[:500], scoring=(2, 3, 5, 2, 4, 1, 0, 5))


def test_cigar_batch_matches_reference_sam(hip, oracle):
    """The same stage against the SAM records the REFERENCE's own host code wrote for the golden read set
    (tests/golden/jobs_golden.npz: POS, CIGAR, NM, MD, strand of every primary alignment)."""
    import torch
    from bwamem_hip import synth
    from bwamem_hip.lib import HostJobs, cigar_batch
    z = np.load(os.path.join(common.GOLDEN, "jobs_golden.npz"))
    g = synth.make_genome(int(z["n_genome"]), seed=int(z["genome_seed"]))
    _, idx = common.genome_and_index(int(z["n_genome"]))
    reads = z["reads"]; n, L = reads.shape
    seeds = {k: z[k] for k in ("rbeg", "qbeg", "score", "n_ref_pos", "prefix")}
    hj = HostJobs(g, reads.reshape(-1), np.arange(n, dtype=np.uint64) * L, np.full(n, L, np.uint32), seeds, n_threads=4)
    out3, _, _ = oracle.extend_batch(*hj.jobs())
    regs = hj.merge(out3)
    sel = []
    for r in range(n):                       # the primary alignment = the best region (duplicates of it are identical)
        idxs = np.nonzero(regs[:, 0] == r)[0]
        if len(idxs) == 0 or z["as_tag"][r] < 0:
            continue
        best = idxs[regs[idxs, 1] == regs[idxs, 1].max()]
        if len(np.unique(regs[best][:, 2:], axis=0)) == 1:
            sel.append(int(best[0]))
    assert len(sel) > 0.95 * n
    dindex = hip.Index.upload(idx, pac=_pack_pac(g), l_pac=len(g))
    rt = _to_dev(torch, synth.codes_to_ascii(reads.reshape(-1)))
    o = (torch.arange(n, dtype=torch.int64) * L).to(torch.int32).cuda()
    l = torch.full((n,), L, dtype=torch.int32).cuda()
    cigar, aln, md = cigar_batch(dindex, rt, o, l, torch.from_numpy(regs.copy()).cuda(), len(sel), sel_t=torch.tensor(sel, dtype=torch.int32).cuda(),
                                 max_cigar=48, md_cap=640)
    torch.cuda.synchronize()
    cigar = cigar.cpu().numpy().view(np.uint32); aln = aln.cpu().numpy(); md = md.cpu().numpy()
    for k, ri in enumerate(sel):
        r = int(regs[ri, 0]); a = aln[k]
        cs = "".join(f"{int(x) >> 4}{'MIDSH'[int(x) & 0xf]}" for x in cigar[k][: a[3]])
        got = (int(np.uint32(a[0])) + 1, cs, int(a[4]), bytes(md[k][: a[6]]).decode(), int(a[2]))
        want = (int(z["sam_pos"][r]), str(z["sam_cigar"][r]), int(z["sam_nm"][r]), str(z["sam_md"][r]), (int(z["sam_flag"][r]) >> 4) & 1)
        assert got == want, (r, got, want)
    hj.free(); dindex.free()


@pytest.mark.parametrize("golden", ["post_golden.npz", "contigs_golden.npz"])
def test_reads_to_sam_text_matches_reference(hip, oracle, golden):
    """The whole chain on the repeat-rich golden read sets (one sequence / three sequences with reads across the cuts): device seeding -> device chaining / jobs -> device extension
    -> device merge -> bmh_finalize_regs (host, like the reference) and bmh_finalize_regs_device (the same on the device) -> bmh_cigar_batch (device) reproduces every SAM record
    the reference's own host code wrote (flag, POS, MAPQ, CIGAR, NM, AS, XS, MD), default run and -a."""
    import ast, ctypes as C, torch
    from bwamem_hip import fmindex, synth
    from bwamem_hip.lib import ChainOpt, ChainWorkspace, PostOpt, cigar_batch, load_library, _np_ptr, _u8p, _u64p, _i32p, _u32p
    z = np.load(os.path.join(common.GOLDEN, golden))
    g = synth.make_genome(int(z["n_genome"]), seed=int(z["genome_seed"]), **ast.literal_eval(str(z["genome_kw"])))
    contigs = ast.literal_eval(str(z["contigs"])) or [("chrS", len(g))]
    c_off = np.ascontiguousarray(np.concatenate([[0], np.cumsum([c[1] for c in contigs])]), dtype=np.int64)
    idx = fmindex.build_fmd_index(g)
    reads = z["reads"]; n, L = reads.shape
    flat = np.ascontiguousarray(reads.reshape(-1))
    pac = _pack_pac(g)
    dindex = hip.Index.upload(idx, pac=pac, l_pac=len(g))
    ws = hip.SeedWorkspace(n, n * L, max_cands=n * L, max_occ=1 << 22)
    r = _to_dev(torch, synth.codes_to_ascii(flat))
    o = (torch.arange(n, dtype=torch.int64) * L).to(torch.int32).cuda()
    l = torch.full((n,), L, dtype=torch.int32).cuda()
    s = ws.seed_batch(dindex, r, o, l, 19)
    cw = ChainWorkspace(n, max(int(s.n_seeds), 1)); cw.set_materialize(False)
    if len(contigs) > 1:
        cw.set_contigs(contigs)
    dj = cw.chain_batch(dindex, r, o, l, s)
    nr = int(dj.n_regs)
    out3 = torch.zeros(max(int(dj.n_jobs), 1), 3, dtype=torch.int32, device="cuda")
    regs = torch.zeros(max(nr, 1), 8, dtype=torch.int32, device="cuda")
    cw.extend(out3); cw.merge(out3, regs)
    torch.cuda.synchronize()
    from bwamem_hip.lib import dev_jobs_to_host
    dh = dev_jobs_to_host(dj, n)                       # per-read region counts and frac_rep of the device job builder
    regs_h = regs.cpu().numpy()[:nr]
    Lb = load_library()
    co = ChainOpt(); Lb.bmh_chain_opt_default(C.byref(co)); ep = hip.ExtParams.default()
    for tag, flag_all in (("def_", 0), ("all_", 1)):
        po = PostOpt(); Lb.bmh_post_opt_default(C.byref(po)); po.flag_all = flag_all
        out = np.zeros((max(nr, 1), 16), np.int32); opr = np.zeros(n, np.uint32)
        fr = np.ascontiguousarray(dh["frac_rep"], dtype=np.float32)
        m = Lb.bmh_finalize_regs(C.byref(co), C.byref(ep), C.byref(po), len(g), _np_ptr(pac, _u8p), n, _np_ptr(flat, _u8p),
                                 _np_ptr(np.arange(n, dtype=np.uint64) * L, _u64p), _np_ptr(np.ascontiguousarray(regs_h), _i32p),
                                 _np_ptr(np.ascontiguousarray(dh["regs_per_read"]), _u32p), fr.ctypes.data_as(C.POINTER(C.c_float)),
                                 len(contigs), c_off.ctypes.data_as(C.c_void_p),
                                 _np_ptr(out, _i32p), _np_ptr(opr, _u32p), 2)
        assert m >= 0
        out = out[:m]
        # the same tail on the device, from the regions where the merge kernel left them: record for record the host's
        from bwamem_hip.lib import finalize_regs_device
        d_out, d_opr = finalize_regs_device(dindex, co, ep, po, r, o, regs, nr, dj.d_regs_per_read, dj.d_frac_rep, n, contigs=contigs)
        assert np.array_equal(d_opr.cpu().numpy().view(np.uint32)[:n], opr) and np.array_equal(d_out.cpu().numpy(), out)
        sel = np.nonzero(out[:, 15])[0].astype(np.int32)
        cigar, aln, md = cigar_batch(dindex, r, o, l, d_out.contiguous(), len(sel), sel_t=torch.from_numpy(sel).cuda(),
                                     max_cigar=48, md_cap=640)
        torch.cuda.synchronize()
        cigar = cigar.cpu().numpy().view(np.uint32); aln = aln.cpu().numpy(); md = md.cpu().numpy()
        got, seen = [], set()
        for k, i in enumerate(sel):
            q = out[i]; a = aln[k]
            cs = "".join(f"{int(x) >> 4}{'MIDSH'[int(x) & 0xf]}" for x in cigar[k][: a[3]])
            if int(q[0]) in seen:
                cs = cs.replace("S", "H")
            seen.add(int(q[0]))
            pos = int(np.uint32(a[0])) | (int(a[1]) << 32)
            rid = int(np.searchsorted(c_off, pos, side="right") - 1)
            got.append((int(q[0]), (16 if a[2] else 0) | int(q[14]), pos - int(c_off[rid]) + 1, int(q[13]), cs, int(a[4]), int(q[1]),
                        int(q[10]) if q[12] < 0 else -1, bytes(md[k][: a[6]]).decode(), contigs[rid][0]))
        for rd in sorted(set(range(n)) - seen):
            got.append((rd, 4, 0, 0, "*", -1, 0, 0, "", "*"))
        got.sort(key=lambda t: t[0])
        rn = [str(x) for x in z[tag + "rname"]]
        want = list(zip(z[tag + "read"].tolist(), z[tag + "flag"].tolist(), z[tag + "pos"].tolist(), z[tag + "mapq"].tolist(), [str(x) for x in z[tag + "cigar"]],
                        z[tag + "nm"].tolist(), z[tag + "as_"].tolist(), z[tag + "xs"].tolist(), [str(x) for x in z[tag + "md"]], rn))
        assert len(got) == len(want), (tag, len(got), len(want))
        bad = [(a, b) for a, b in zip(got, want) if a != b]
        assert not bad, (tag, len(bad), bad[:3])
    # ... and the SAM text itself, byte for byte (default run): DEVICE finalize -> device CIGAR of every record the formatter
    # needs (reported ones and XA candidates) -> bmh_format_sam: nothing between the reads and the records leaves HBM
    from bwamem_hip.lib import format_sam
    po = PostOpt(); Lb.bmh_post_opt_default(C.byref(po))
    d_out, d_opr = finalize_regs_device(dindex, co, ep, po, r, o, regs, nr, dj.d_regs_per_read, dj.d_frac_rep, n, contigs=contigs)
    out = np.ascontiguousarray(d_out.cpu().numpy()); opr = np.ascontiguousarray(d_opr.cpu().numpy().view(np.uint32)[:n]); m = len(out)
    need = np.zeros(max(m, 1), np.uint8)
    k = Lb.bmh_sam_need_cigar(C.byref(po), _np_ptr(out, _i32p), _np_ptr(opr, _u32p), n, _np_ptr(need, _u8p))
    sel = np.nonzero(need[:m])[0].astype(np.int32)
    assert k == len(sel)
    cigar, aln, md = cigar_batch(dindex, r, o, l, torch.from_numpy(out.copy()).cuda(), len(sel), sel_t=torch.from_numpy(sel).cuda(), max_cigar=48, md_cap=640)
    torch.cuda.synchronize()
    slot = np.full(max(m, 1), -1, np.int64); slot[sel] = np.arange(len(sel))
    # the selection and the packing the native pipeline does on the device (csrc/sam_kernels.hip): the host's list, the fixed slots' contents
    from bwamem_hip.lib import sam_select_device, cigar_pack
    for fa in (0, 1):
        po2 = PostOpt(); Lb.bmh_post_opt_default(C.byref(po2)); po2.flag_all = fa
        need2 = np.zeros(max(m, 1), np.uint8)
        Lb.bmh_sam_need_cigar(C.byref(po2), _np_ptr(out, _i32p), _np_ptr(opr, _u32p), n, _np_ptr(need2, _u8p))
        d_sel, d_slot = sam_select_device(po2, d_out.contiguous(), d_opr[:n].contiguous())
        want_sel = np.nonzero(need2[:m])[0]
        assert np.array_equal(d_sel.cpu().numpy(), want_sel), fa
        want_slot = np.full(m, -1, np.int64); want_slot[want_sel] = np.arange(len(want_sel))
        assert np.array_equal(d_slot.cpu().numpy(), want_slot), fa
    for mc, mdc in ((48, 640), (4, 4)):                      # (small slots: some alignments overflow them and take no words)
        cg2, al2, md2 = cigar_batch(dindex, r, o, l, d_out.contiguous(), len(sel), sel_t=torch.from_numpy(sel).cuda(), max_cigar=mc, md_cap=mdc)
        off_t, packed_t = cigar_pack(al2, cg2, md2)
        offp, pk = off_t.cpu().numpy().astype(np.int64), packed_t.cpu().numpy().view(np.uint32)
        a2, c2, m2 = al2.cpu().numpy(), cg2.cpu().numpy().view(np.uint32), md2.cpu().numpy()
        n_over = 0
        for k2 in range(len(sel)):
            if a2[k2, 7] & ~2:
                assert offp[k2 + 1] == offp[k2]; n_over += 1
                continue
            nc, ml = int(a2[k2, 3]), int(a2[k2, 6])
            assert offp[k2 + 1] - offp[k2] == nc + (ml + 4) // 4
            assert np.array_equal(pk[offp[k2]: offp[k2] + nc], c2[k2, :nc])
            assert pk[offp[k2] + nc: offp[k2 + 1]].tobytes()[: ml + 1] == m2[k2, : ml + 1].tobytes()
        assert offp[len(sel)] == len(pk) and (mc == 48 or n_over > 0)
    txt = format_sam(po, [f"r{i}" for i in range(n)], flat, np.arange(n, dtype=np.uint64) * L, np.full(n, L, np.uint32), contigs, out, opr, slot,
                     aln.cpu().numpy(), cigar.cpu().numpy().view(np.uint32), md.cpu().numpy())
    want = bytes(z["sam_text"]).decode()
    if txt != want:
        gl, wl = txt.split("\n"), want.split("\n")
        assert False, (len(gl), len(wl), [(a, b) for a, b in zip(gl, wl) if a != b][:2])
    # ... and the same text written by the device (bmh_sam_text_sizes / _write) from the device's own selection and packed CIGARs
    from bwamem_hip.lib import sam_text_device
    d_sel, d_slot = sam_select_device(po, d_out.contiguous(), d_opr[:n].contiguous())
    cg3, al3, md3 = cigar_batch(dindex, r, o, l, d_out.contiguous(), int(d_sel.shape[0]), sel_t=d_sel.contiguous(), max_cigar=48, md_cap=640)
    off3, pk3 = cigar_pack(al3, cg3, md3)
    txt_dev = sam_text_device(po, [f"r{i}" for i in range(n)], r, o, l, contigs, d_out.contiguous(), d_opr[:n].contiguous(), d_slot.contiguous(), al3, off3.contiguous(), pk3.contiguous())
    assert txt_dev.decode() == want
    cw.free(); ws.free(); dindex.free()


@pytest.mark.parametrize("golden", ["post_golden.npz", "contigs_golden.npz", "pe_golden.npz", "pe_contigs_golden.npz", "alt_golden.npz", "pe_alt_golden.npz", "pe_alt2_golden.npz"])
def test_aligner_writes_reference_sam(hip, tmp_path, golden):
    """bwamem_hip.aligner (index files + FASTA -> SAM over the device-resident path) against the SAM text recorded from the
    reference binary: single-end (repeat-rich, three sequences) and interleaved paired-end (-p); alt_*: a genome with ALT contigs
    named in <prefix>.alt (chain filter, two-round primary marking, MAPQ, XA / pa tags, soft clips on ALT hits); pe_alt2: an ALT contig with a stretch the
    primary assembly lacks and pairs whose mates reach it with nothing but a weak hit on the primary assembly (mem_sam_pe shows the mate the best ALT hit,
    src/bwamem_pair.c:376-389)."""
    import ast, io
    from bwamem_hip import fmindex, synth
    from bwamem_hip.aligner import Aligner
    z = np.load(os.path.join(common.GOLDEN, golden))
    if "genome_packed" in z.files:                     # (the ALT genome is not a plain generator call: it travels with the golden file)
        n = int(z["n_genome"]); bits = np.unpackbits(z["genome_packed"])[: 2 * n].reshape(n, 2)
        g = (bits[:, 0] * 2 + bits[:, 1]).astype(np.uint8)
    else:
        g = synth.make_genome(int(z["n_genome"]), seed=int(z["genome_seed"]), **ast.literal_eval(str(z["genome_kw"])))
    contigs = ast.literal_eval(str(z["contigs"])) if "contigs" in z.files else None
    prefix = str(tmp_path / "g.fa")
    fmindex.write_index(prefix, fmindex.build_fmd_index(g)); fmindex.write_bns(prefix, g, contigs=contigs)
    if "alt_file" in z.files:
        with open(prefix + ".alt", "wb") as f:
            f.write(bytes(z["alt_file"]))
    reads = z["reads"]
    pe = golden.startswith("pe_")
    fq = str(tmp_path / "r.fa")
    asc = synth.codes_to_ascii(reads)
    with open(fq, "wb") as f:
        for i in range(len(asc)):
            f.write((b">p%d\n" % (i // 2)) if pe else (b">r%d\n" % i)); f.write(asc[i].tobytes()); f.write(b"\n")
    al = Aligner(prefix, n_threads=2)
    buf = io.StringIO()
    al.align_file(fq, buf, batch_reads=1 << 30 if pe else 256, paired=pe)       # single-end: several batches
    body = "".join(l + "\n" for l in buf.getvalue().split("\n") if l and l[0] != "@")
    want = bytes(z["sam_text"]).decode()
    if body != want:
        gl, wl = body.split("\n"), want.split("\n")
        assert False, (len(gl), len(wl), [(a, b) for a, b in zip(gl, wl) if a != b][:2])
    assert buf.getvalue().startswith(bytes(z["sam_header"]).decode())
    # the native pipeline wrote that text on the device (bmh_sam_text_*); the host formatter from records copied home, and the host's selection
    # of the records that need a CIGAR, give the same bytes -- and so does a batch whose region tail ran on the host (device tail refused)
    for env, val in (("BMH_ALIGNER_HOST_FORMAT", "1"), ("BMH_ALIGNER_HOST_SELECT", "1"), ("BMH_FIN_FORCE_ECAPACITY", "1"), ("BMH_ALIGNER_PE_HOST_DEDUP", "1"), ("BMH_ALIGNER_PE_HOST", "1"),
                     ("BMH_ALIGNER_ALT_HOST_PATCH", "1"),      # (ALT contigs: the device tail without the table, the reads that touch an ALT contig redone on the host)
                     ("BMH_ALIGNER_STREAM", "0")):          # (STREAM=0: the whole file loaded first, bmh_aligner_run on its cuts, instead of bmh_aligner_run_fasta)
        os.environ[env] = val
        try:
            buf2 = io.StringIO()
            al.align_file(fq, buf2, batch_reads=1 << 30 if pe else 256, paired=pe)
            assert buf2.getvalue() == buf.getvalue(), env
        finally:
            del os.environ[env]
    if not pe:
        # batches that grow and shrink: the aligner keeps its workspaces and pinned buffers between batches and replaces them when a batch
        # needs more; the text comes back as a view of the library's buffer (binary output)
        from bwamem_hip.aligner import read_fasta_reads
        rs = read_fasta_reads(fq)
        n = len(rs)
        cuts = [0, 50, min(700, n - 20), min(710, n - 10), n]
        parts = [bytes(al.align_batch(rs.slice(b, e), id0=b, as_bytes="view")) for b, e in zip(cuts[:-1], cuts[1:]) if e > b]
        assert b"".join(parts).decode() == want
        # a batch the device tail refuses (BMH_ECAPACITY: a read beyond its fixed limits) takes the host tail instead of failing the run
        os.environ["BMH_FIN_FORCE_ECAPACITY"] = "1"
        try:
            before = getattr(al, "host_tail_batches", 0)
            assert bytes(al.align_batch(rs.slice(0, n), id0=0, as_bytes="view")).decode() == want
            assert al.has_alt or al.host_tail_batches == before + 1          # (an index with ALT contigs takes the host tail in the first place)
        finally:
            del os.environ["BMH_FIN_FORCE_ECAPACITY"]
    al.close()
    # ... and the reference's -a run (every hit a record: secondary ones included) record by record -- flag, POS, MAPQ, CIGAR, NM, AS, XS, MD, RNAME --
    # through the same device path with flag_all set
    if "all_read" in z.files:
        al2 = Aligner(prefix, n_threads=2)
        al2.po.flag_all = 1
        buf3 = io.StringIO()
        al2.align_file(fq, buf3, batch_reads=1 << 30 if pe else 256, paired=pe)
        got = []
        for line in buf3.getvalue().split("\n"):
            if not line or line[0] == "@":
                continue
            c = line.split("\t")
            tags = {t[:2]: t[5:] for t in c[11:]}
            rd = int(c[0][1:]) if not pe else 2 * int(c[0][1:]) + (1 if int(c[1]) & 0x80 else 0)
            got.append((rd, int(c[1]), int(c[3]), int(c[4]), c[5], int(tags.get("NM", -1)), int(tags.get("AS", -1)), int(tags.get("XS", -1)), tags.get("MD", ""), c[2]))
        want_a = list(zip(z["all_read"].tolist(), z["all_flag"].tolist(), z["all_pos"].tolist(), z["all_mapq"].tolist(), [str(x) for x in z["all_cigar"]],
                          z["all_nm"].tolist(), z["all_as_"].tolist(), z["all_xs"].tolist(), [str(x) for x in z["all_md"]], [str(x) for x in z["all_rname"]]))
        assert len(got) == len(want_a), (len(got), len(want_a))
        bad = [(a, b) for a, b in zip(got, want_a) if a != b]
        assert not bad, (len(bad), bad[:3])
        al2.close()


@pytest.mark.parametrize("pe", [False, True])
def test_native_pipeline_device_text_equals_host_text(hip, tmp_path, pe):
    """bmh_aligner_run on a repeat-rich genome of three sequences, reads with many substitutions and indels among them (alignments that overflow
    the fixed CIGAR / MD slots and are redone on the device): the text written on the device (bmh_sam_select_device, bmh_cigar_pack,
    bmh_sam_text_*) is the text of the host path (records to the host, bmh_sam_need_cigar, bmh_format_sam), byte for byte."""
    import io
    from bwamem_hip import fmindex, synth
    from bwamem_hip.aligner import Aligner
    g = synth.make_genome(1_500_000, seed=11, repeat_frac=0.3)
    contigs = [("chrA", 600_000), ("chrB", 500_000), ("chrC", 400_000)]
    prefix = str(tmp_path / "g.fa")
    fmindex.write_index(prefix, fmindex.build_fmd_index(g)); fmindex.write_bns(prefix, g, contigs=contigs)
    n = 12000
    L = 250
    reads = (synth.make_pairs(g, n // 2, L, seed=5) if pe else synth.make_reads(g, n, L, seed=5))[0].copy()
    rng = np.random.default_rng(9)
    for r in rng.choice(n, 600, replace=False):                   # many substitutions (an MD string beyond 95 characters) ...
        pos = rng.choice(L, 45, replace=False)
        reads[r, pos] = (reads[r, pos] + rng.integers(1, 4, len(pos))) % 4
    asc = [a.tobytes() for a in synth.codes_to_ascii(reads)]
    for r in rng.choice(n, 600, replace=False):                   # ... and a dozen short indels (more operations than the fixed slot holds)
        b = bytearray(asc[r])
        for p in sorted(rng.choice(np.arange(15, L - 15, 16), 10, replace=False).tolist(), reverse=True):
            if rng.integers(2):
                del b[p]
            else:
                b.insert(p, b"ACGT"[int(rng.integers(4))])
        asc[r] = bytes(b)
    fq = str(tmp_path / "r.fa")
    with open(fq, "wb") as f:
        for i, a in enumerate(asc):
            f.write((b">p%d\n" % (i // 2)) if pe else (b">r%d\n" % i)); f.write(a); f.write(b"\n")
    al = Aligner(prefix, n_threads=4)
    texts = {}
    import ctypes as C
    lib = hip.load_library()
    chk0 = (C.c_uint64 * 5)(); chk1 = (C.c_uint64 * 5)()
    lib.bmh_rescue_check_counts(chk0)
    # (BMH_ALIGNER_RESCUE_DEV=1: the rescue's windows found by rescue_jobs_kernel instead of the host's first walk; RESCUE_CHECK runs that walk beside it)
    for env in ("", "BMH_ALIGNER_HOST_FORMAT", "BMH_ALIGNER_HOST_SELECT") + (("BMH_ALIGNER_PE_HOST_DEDUP", "BMH_ALIGNER_RESCUE_DEV") if pe else ()):
        names = [env] if env else []
        if env == "BMH_ALIGNER_RESCUE_DEV":
            names.append("BMH_RESCUE_CHECK")
        for k in names:
            os.environ[k] = "1"
        try:
            buf = io.BytesIO()
            al.align_file(fq, buf, batch_reads=5000, paired=pe)
            texts[env] = buf.getvalue()
        finally:
            for k in names:
                del os.environ[k]
    lib.bmh_rescue_check_counts(chk1)
    if pe:      # batches checked, jobs found on the device, no pair whose `active` flag differs from the host walk's
        d = [int(b) - int(a) for a, b in zip(chk0, chk1)]
        assert d[0] >= 3 and d[1] == n // 2 and d[2] > 100 and d[3] == 0, d
        print("rescue windows on the device vs the host's first walk: batches, pairs, jobs, active differs, call list differs:", d)
    body = texts[""]
    assert body.count(b"\n") >= n and b"\tXA:Z:" in body and b"\tSA:Z:" in body
    lines = [l for l in body.split(b"\n") if l and not l.startswith(b"@")]
    assert max(l.split(b"\t")[5].count(b"I") + l.split(b"\t")[5].count(b"D") for l in lines) >= 8       # the overflow path was taken
    for env in [e for e in texts if e]:                            # (BMH_ALIGNER_PE_HOST_DEDUP: mem_sort_dedup_patch of the pairs on host threads instead of the device)
        if texts[env] != body:
            a, b = body.split(b"\n"), texts[env].split(b"\n")
            assert False, (env, len(a), len(b), [(x, y) for x, y in zip(a, b) if x != y][:2])
    al.close()


@pytest.mark.parametrize("opts", [{}, {"flag_all": 1}, {"no_multi": 1, "softclip": 1}], ids=["default", "all", "M_Y"])
@pytest.mark.parametrize("pe", [False, True])
def test_native_pipeline_with_alt_contigs_device_forms_equal_host_forms(hip, tmp_path, pe, opts):
    """An index with ALT contigs (<prefix>.alt) through bmh_aligner_run: the device tail's ALT rules and, for pairs, pair_kernel (pairs without a hit on an ALT
    contig) + the host's mem_sam_pe (the others), against the host forms: the host tail of the reads that touch an ALT contig (BMH_ALIGNER_ALT_HOST_PATCH), all
    of mem_sam_pe on the host (BMH_ALIGNER_PE_HOST), the host's formatter, and the Python loop (host tail of every read)."""
    import io
    from bwamem_hip import fmindex, synth
    from bwamem_hip.aligner import Aligner
    g = synth.make_genome(1_500_000, seed=12, repeat_frac=0.4, repeat_len=(150, 600), repeat_copies=(4, 60), repeat_div=0.02)
    contigs = [("chrA", 600_000), ("chrB", 500_000), ("chrB_alt1", 250_000), ("chrA_alt1", 150_000)]
    prefix = str(tmp_path / "g.fa")
    fmindex.write_index(prefix, fmindex.build_fmd_index(g)); fmindex.write_bns(prefix, g, contigs=contigs)
    with open(prefix + ".alt", "w") as f:
        f.write("@SQ\tSN:chrB_alt1\nchrB_alt1\t0\tchrB\t1\t60\t100M\nchrA_alt1\t0\tchrA\t1\t60\t100M\n")
    n, L = 20000, 150
    reads = (synth.make_pairs(g, n // 2, L, seed=6) if pe else synth.make_reads(g, n, L, seed=6))[0]
    fq = str(tmp_path / "r.fa")
    with open(fq, "wb") as f:
        for i, a in enumerate(synth.codes_to_ascii(reads)):
            f.write((b">p%d\n" % (i // 2)) if pe else (b">r%d\n" % i)); f.write(a.tobytes()); f.write(b"\n")
    al = Aligner(prefix, n_threads=4)
    assert al.has_alt
    for k_, v_ in opts.items():                                    # (-a: every hit a record; -M -Y: shorter split hits secondary, soft clips everywhere)
        setattr(al.po, k_, v_)
    texts = {}
    for env in ("", "BMH_ALIGNER_HOST_FORMAT", "BMH_ALIGNER_PE_HOST" if pe else "BMH_ALIGNER_ALT_HOST_PATCH", "BMH_ALIGNER_NATIVE") + (("BMH_ALIGNER_RESCUE_DEV",) if pe else ()):
        if env:
            os.environ[env] = "0" if env == "BMH_ALIGNER_NATIVE" else "1"
        try:
            buf = io.BytesIO()
            al.align_file(fq, buf, batch_reads=8000, paired=pe)
            texts[env] = buf.getvalue()
        finally:
            if env:
                del os.environ[env]
    body = texts[""]
    lines = [l.split(b"\t") for l in body.split(b"\n") if l and not l.startswith(b"@")]
    on_alt = sum(1 for l in lines if l[2].endswith(b"_alt1"))
    assert len(lines) >= n and on_alt > 500 and body.count(b"\tpa:f:") > 100 and (b"\tXA:Z:" in body or opts.get("flag_all")), (len(lines), on_alt, body.count(b"\tpa:f:"))
    for env in [e for e in texts if e]:
        if texts[env] != body:
            a, b = body.split(b"\n"), texts[env].split(b"\n")
            assert False, (env, len(a), len(b), [(x, y) for x, y in zip(a, b) if x != y][:2])
    al.close()


@pytest.mark.parametrize("pe", [False, True])
def test_read_group_tag_on_every_record(hip, tmp_path, pe):
    """-R: the @RG line behind the @SQ lines, RG:Z:<id> on every record -- unmapped reads, secondary and supplementary records too -- behind AS / XS and in front of
    SA (mem_aln2sam, src/bwamem.c:1631-1634), from the device's SAM writer and from the host's formatter: the same bytes.  (Against the reference binary: scripts/e2e_wide.sh.)"""
    import io
    from bwamem_hip import fmindex, synth
    from bwamem_hip.aligner import Aligner
    g = synth.make_genome(800_000, seed=15, repeat_frac=0.4, repeat_len=(150, 600), repeat_copies=(4, 60), repeat_div=0.02)
    prefix = str(tmp_path / "g.fa")
    fmindex.write_index(prefix, fmindex.build_fmd_index(g)); fmindex.write_bns(prefix, g, contigs=[("chrA", 500_000), ("chrB", 300_000)])
    n, L = 6000, 150
    reads = (synth.make_pairs(g, n // 2, L, seed=6) if pe else synth.make_reads(g, n, L, seed=6))[0].copy()
    reads[7] = np.random.default_rng(1).integers(0, 4, L)            # an unalignable read
    for r in range(11, n, 97):                                        # chimeric reads: supplementary records and SA tags
        p2 = (r * 7919) % (len(g) - L); reads[r, 70:] = g[p2:p2 + L - 70]
    fq = str(tmp_path / "r.fa")
    with open(fq, "wb") as f:
        for i, a in enumerate(synth.codes_to_ascii(reads)):
            f.write((b">p%d\n" % (i // 2)) if pe else (b">r%d\n" % i)); f.write(a.tobytes()); f.write(b"\n")
    al = Aligner(prefix, n_threads=4)
    al.set_options(["-a", "-R", "@RG\\tID:g7.x\\tSM:s1"])
    texts = {}
    for env in ("", "BMH_ALIGNER_HOST_FORMAT", "BMH_ALIGNER_NATIVE"):
        if env:
            os.environ[env] = "0" if env == "BMH_ALIGNER_NATIVE" else "1"
        try:
            buf = io.BytesIO(); al.align_file(fq, buf, batch_reads=2500 if not pe else 1 << 30, paired=pe); texts[env] = buf.getvalue()
        finally:
            if env:
                del os.environ[env]
    body = texts[""]
    hdr = [l for l in body.split(b"\n") if l.startswith(b"@")]
    assert hdr == [b"@SQ\tSN:chrA\tLN:500000", b"@SQ\tSN:chrB\tLN:300000", b"@RG\tID:g7.x\tSM:s1"], hdr
    recs = [l for l in body.split(b"\n") if l and not l.startswith(b"@")]
    assert len(recs) > n and all(b"\tRG:Z:g7.x" in l for l in recs)
    assert any(int(l.split(b"\t")[1]) & 4 for l in recs) and any(b"\tSA:Z:" in l for l in recs) and any(int(l.split(b"\t")[1]) & 0x100 for l in recs)
    for l in recs:                                                    # behind XS / AS, in front of SA and XA
        t = [x[:2] for x in l.split(b"\t")[11:]]
        k = t.index(b"RG")
        assert t[k - 1] in (b"XS", b"AS") and all(x in (b"SA", b"XA", b"pa") for x in t[k + 1:]), l
    for env, t in texts.items():
        assert t == body, env
    al.close()


@pytest.mark.parametrize("flag_all", [0, 1])
def test_device_tail_with_alt_contigs_equals_host_tail(hip, flag_all):
    """bmh_finalize_regs_device WITH an ALT table (second marking round, secondary_all, alt_sc, mem_reg2sam's rules for ALT hits -- lane form and the wave classes)
    against bmh_finalize_regs with the same table: 40 000 reads of a repeat-rich genome of five sequences two of which are flagged, among them reads with hundreds
    of regions."""
    import ctypes as C, torch
    from bwamem_hip import fmindex, synth
    from bwamem_hip.lib import ChainOpt, ChainWorkspace, PostOpt, load_library, finalize_regs_device, dev_jobs_to_host, _np_ptr, _u8p, _u64p, _i32p, _u32p
    Lb = load_library()
    g = synth.make_genome(3_000_000, seed=31, repeat_frac=0.5, repeat_len=(150, 700), repeat_copies=(20, 600), repeat_div=0.02)
    contigs = [("c1", 1_000_000), ("c2", 900_000), ("c3", 500_000), ("c2_alt", 400_000), ("c3_alt", 200_000)]
    is_alt = np.ascontiguousarray([0, 0, 0, 1, 1], dtype=np.uint8)
    c_off = np.ascontiguousarray(np.concatenate([[0], np.cumsum([c[1] for c in contigs])]), dtype=np.int64)
    idx = fmindex.build_fmd_index(g, device="cuda:0")
    n, L = 40_000, 150
    reads = synth.make_reads(g, n, L, seed=3)[0]
    flat = np.ascontiguousarray(reads.reshape(-1))
    pac = _pack_pac(g)
    dindex = hip.Index.upload(idx, pac=pac, l_pac=len(g))
    ws = hip.SeedWorkspace(n, n * L, max_cands=n * L, max_occ=1 << 22)
    r = _to_dev(torch, synth.codes_to_ascii(flat))
    o = (torch.arange(n, dtype=torch.int64) * L).to(torch.int32).cuda()
    l = torch.full((n,), L, dtype=torch.int32).cuda()
    s = ws.seed_batch(dindex, r, o, l, 19)
    cw = ChainWorkspace(n, max(int(s.n_seeds), 1)); cw.set_materialize(False); cw.set_contigs(contigs); cw.set_alt(is_alt)
    dj = cw.chain_batch(dindex, r, o, l, s)
    nr = int(dj.n_regs)
    out3 = torch.zeros(max(int(dj.n_jobs), 1), 3, dtype=torch.int32, device="cuda"); regs = torch.zeros(max(nr, 1), 8, dtype=torch.int32, device="cuda")
    cw.extend(out3); cw.merge(out3, regs); torch.cuda.synchronize()
    dh = dev_jobs_to_host(dj, n)
    co = ChainOpt(); Lb.bmh_chain_opt_default(C.byref(co)); co.contig_is_alt = is_alt.ctypes.data_as(C.c_void_p).value
    ep = hip.ExtParams.default()
    po = PostOpt(); Lb.bmh_post_opt_default(C.byref(po)); po.flag_all = flag_all; po.id0 = 1000; po.contig_is_alt = is_alt.ctypes.data_as(C.c_void_p).value
    out = np.zeros((max(nr, 1), 16), np.int32); opr = np.zeros(n, np.uint32)
    fr = np.ascontiguousarray(dh["frac_rep"], dtype=np.float32)
    m = Lb.bmh_finalize_regs(C.byref(co), C.byref(ep), C.byref(po), len(g), _np_ptr(pac, _u8p), n, _np_ptr(flat, _u8p), _np_ptr(np.arange(n, dtype=np.uint64) * L, _u64p),
                             _np_ptr(np.ascontiguousarray(regs.cpu().numpy()[:nr]), _i32p), _np_ptr(np.ascontiguousarray(dh["regs_per_read"]), _u32p), fr.ctypes.data_as(C.POINTER(C.c_float)),
                             len(contigs), c_off.ctypes.data_as(C.c_void_p), _np_ptr(out, _i32p), _np_ptr(opr, _u32p), 8)
    assert m >= 0
    out = out[:m]
    d_out, d_opr = finalize_regs_device(dindex, co, ep, po, r, o, regs, nr, dj.d_regs_per_read, dj.d_frac_rep, n, contigs=contigs)
    got = d_out.cpu().numpy()
    assert np.array_equal(d_opr.cpu().numpy().view(np.uint32)[:n], opr)
    if not np.array_equal(got, out):
        bad = np.nonzero((got != out).any(1))[0]
        assert False, (len(bad), bad[:5], out[bad[:3]], got[bad[:3]])
    assert (out[:, 15] & 2).sum() > 2000 and (out[:, 15] >> 2 > 0).sum() > 200 and (out[:, 12] == 0x7FFFFFFF).sum() > 500 and opr.max() > 100
    cw.free(); ws.free(); dindex.free()


def test_device_text_pa_tag_rounds_like_printf(hip):
    """The pa:f tag of the device formatter (score / score of the shadowing ALT hit as %.3f, src/bwamem.c:1663) for every pair of scores up to 300: the
    decimal string C's printf writes (exact value of the nearest double, ties to even: 1/16 -> 0.062, 3/16 -> 0.188)."""
    import ctypes as C, torch
    from bwamem_hip.lib import PostOpt, load_library, sam_text_device
    Lb = load_library()
    po = PostOpt(); Lb.bmh_post_opt_default(C.byref(po))
    A, Bm = np.meshgrid(np.arange(1, 301), np.arange(1, 301), indexing="ij")
    a, b = A.reshape(-1).astype(np.int32), Bm.reshape(-1).astype(np.int32)
    n, L = len(a), 10
    fin = np.zeros((n, 16), np.int32)
    fin[:, 0] = np.arange(n); fin[:, 1] = a; fin[:, 3] = L; fin[:, 6] = L; fin[:, 8] = a; fin[:, 9] = 100; fin[:, 10] = -1; fin[:, 12] = -1; fin[:, 13] = 60
    fin[:, 15] = 1 | (b << 2)
    aln = np.zeros((n, 8), np.int32); aln[:, 3] = 1; aln[:, 6] = 2                      # position 0, forward, one operation, NM 0, MD "10"
    packed = np.zeros((n, 2), np.uint32); packed[:, 0] = L << 4; packed[:, 1] = np.frombuffer(b"10\0\0", np.uint32)[0]
    dev = "cuda"
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    reads = t(np.frombuffer(b"ACGTACGTAC" * n, np.uint8).copy())
    txt = sam_text_device(po, [f"r{i}" for i in range(n)], reads, t((np.arange(n) * L).astype(np.int32)), t(np.full(n, L, np.int32)), [("c", 1000)],
                          t(fin), t(np.ones(n, np.int32)), t(np.arange(n, dtype=np.int32)), t(aln), t((np.arange(n + 1) * 2).astype(np.int32)), t(packed.reshape(-1).view(np.int32)))
    got = [l.split(b"pa:f:")[1].split(b"\t")[0].decode() for l in txt.split(b"\n") if l]
    want = ["%.3f" % (float(x) / float(y)) for x, y in zip(a.tolist(), b.tolist())]
    assert len(got) == n
    bad = [(int(x), int(y), g, w) for x, y, g, w in zip(a, b, got, want) if g != w]
    assert not bad, (len(bad), bad[:5])
    assert "0.062" in got and "0.188" in got


@pytest.mark.parametrize("pe", [False, True])
def test_read_file_batch_by_batch_equals_the_loaded_file(hip, tmp_path, pe):
    """bmh_aligner_run_fasta (a loader thread cuts and fills the batches of the mapped file while the lanes work) against bmh_reads_load_fasta + bmh_aligner_run on
    the same cuts: by read count and by bases, batches of a handful of reads and of thousands, CR LF line ends, blank lines, a last line without a newline, header
    lines with descriptions."""
    import io
    from bwamem_hip import fmindex, synth
    from bwamem_hip.aligner import Aligner
    g = synth.make_genome(400_000, seed=8, repeat_frac=0.25)
    prefix = str(tmp_path / "g.fa")
    fmindex.write_index(prefix, fmindex.build_fmd_index(g)); fmindex.write_bns(prefix, g, contigs=[("a", 150_000), ("b", 250_000)])
    n = 5000
    rng = np.random.default_rng(12)
    if pe:
        reads = synth.make_pairs(g, n // 2, 101, seed=2)[0]
        seqs = [synth.codes_to_ascii(r).tobytes() for r in reads]
    else:
        seqs = [synth.codes_to_ascii(g[p:p + int(l)]).tobytes() for p, l in zip(rng.integers(0, len(g) - 260, n), rng.integers(30, 251, n))]     # ragged lengths
    fq = str(tmp_path / "r.fa")
    with open(fq, "wb") as f:
        for i, b in enumerate(seqs):
            nl = b"\r\n" if i % 5 == 0 else b"\n"
            f.write(b">" + ((b"p%d" % (i // 2)) if pe else (b"r%d" % i)) + (b" a description" if i % 3 == 0 else b"") + nl)
            if i % 97 == 0:
                f.write(nl)
            f.write(b + (b"" if i == n - 1 else nl))
    al = Aligner(prefix, n_threads=2)
    for kw in (dict(batch_reads=7 if not pe else 8), dict(batch_reads=1000), dict(chunk_bases=20_000), dict(chunk_bases=300_000), dict(batch_reads=1 << 30)):
        texts = []
        for stream in ("1", "0"):
            os.environ["BMH_ALIGNER_STREAM"] = stream
            try:
                buf = io.BytesIO()
                assert al.align_file(fq, buf, paired=pe, **kw) == n
                texts.append(buf.getvalue())
            finally:
                del os.environ["BMH_ALIGNER_STREAM"]
        assert texts[0].count(b"\n") >= n
        if texts[0] != texts[1]:
            a, b = texts[0].split(b"\n"), texts[1].split(b"\n")
            assert False, (kw, len(a), len(b), [(x, y) for x, y in zip(a, b) if x != y][:2])
    # a file whose lines do not alternate is refused, whichever batch the fault lies in
    bad = str(tmp_path / "bad.fa")
    with open(bad, "wb") as f:
        f.write(b">x0\nACGTACGTACGTACGTACGTACGTACGT\n" * 50 + b">x\n>y\nACGT\n")
    with pytest.raises(RuntimeError):
        al.align_file(bad, io.BytesIO(), batch_reads=10)
    al.close()


@pytest.mark.parametrize("pe", [False, True])
def test_native_pipeline_edge_reads(hip, tmp_path, pe):
    """Reads the device forms must not trip over: all-N reads, reads shorter than a seed, lower-case letters, IUPAC codes, a one-read batch, pairs with one or both
    mates unmappable, names of different lengths: device text == host-formatter text == the batch-after-batch Python loop's text."""
    import io
    from bwamem_hip import fmindex, synth
    from bwamem_hip.aligner import Aligner
    g = synth.make_genome(300_000, seed=4, repeat_frac=0.2)
    prefix = str(tmp_path / "g.fa")
    fmindex.write_index(prefix, fmindex.build_fmd_index(g)); fmindex.write_bns(prefix, g, contigs=[("c1", 100_000), ("chrTwo_long_name", 200_000)])
    rng = np.random.default_rng(3)
    L = 120
    base = [synth.codes_to_ascii(g[p:p + L]).tobytes() for p in rng.integers(0, len(g) - L, 40)]
    seqs = []
    for i, b in enumerate(base):
        if i % 8 == 1: b = b"N" * L
        elif i % 8 == 2: b = b[:12]
        elif i % 8 == 3: b = b.lower()
        elif i % 8 == 4: b = b[:30] + b"RYKM" + b[34:]
        elif i % 8 == 5: b = bytes(rng.choice(list(b"ACGT"), L).tolist())          # random: no hit
        elif i % 8 == 6: b = b[:60] + b"NNNNN" + b[65:]
        seqs.append(b)
    fq = str(tmp_path / "r.fa")
    with open(fq, "wb") as f:
        for i, b in enumerate(seqs):
            name = (b"pair%d" % (i // 2)) if pe else (b"r%d" % i if i % 3 else b"read_with_a_longer_name_%d" % i)
            f.write(b">" + name + b"\n" + b + b"\n")
    al = Aligner(prefix, n_threads=2)
    texts = {}
    for env, batch in (("", 40), ("", 1 if not pe else 2), ("BMH_ALIGNER_HOST_FORMAT", 40), ("BMH_ALIGNER_NATIVE=0", 40)) + ((("BMH_ALIGNER_PE_HOST", 40), ("BMH_ALIGNER_RESCUE_DEV", 40)) if pe else ()):
        key, val = (env.split("=") + ["1"])[:2] if env else ("", "")
        if key:
            os.environ[key] = val
        try:
            buf = io.BytesIO()
            al.align_file(fq, buf, batch_reads=batch, paired=pe)
            texts[(env, batch)] = buf.getvalue()
        finally:
            if key:
                del os.environ[key]
    body = texts[("", 40)]
    assert body.count(b"\n") >= len(seqs) and b"\t4\t*\t0\t0\t*" in body or pe
    for k, t in texts.items():
        if pe and k == ("", 2):
            continue                                              # (pairs: another batch size means other insert-size statistics)
        if t != body:
            a, b = body.split(b"\n"), t.split(b"\n")
            assert False, (k, len(a), len(b), [(x, y) for x, y in zip(a, b) if x != y][:2])
    al.close()


def test_aligner_refuses_flanks_beyond_the_extension_kernels(hip, tmp_path):
    """Reads of 1000 bp seeded near one end need a query side longer than the 768 bases the DP kernels take: the aligner must
    raise instead of folding the kernels' INT32_MIN placeholders into regions and SAM (and still aligns 700 bp reads)."""
    from bwamem_hip import fmindex, synth
    from bwamem_hip.aligner import Aligner
    g = synth.make_genome(300_000, seed=3)
    prefix = str(tmp_path / "g.fa")
    fmindex.write_index(prefix, fmindex.build_fmd_index(g)); fmindex.write_bns(prefix, g)
    rng = np.random.default_rng(2)

    def reads_of(ln, n):
        rows = []
        for _ in range(n):
            p = int(rng.integers(0, len(g) - ln))
            x = g[p:p + ln].copy()
            x[40:ln:23] = (x[40:ln:23] + 1) & 3            # exact only over the first 40 bases: the right flank is ~ln - 40 long
            rows.append(synth.codes_to_ascii(x))
        return rows
    al = Aligner(prefix, n_threads=2)
    names = [f"r{i}" for i in range(8)]
    txt = al.align_batch(names, reads_of(700, 8))
    assert txt.count("\n") >= 8 and "\t0\t" in txt or "\t16\t" in txt
    with pytest.raises(NotImplementedError, match="768"):
        al.align_batch(names, reads_of(1000, 8))
    al.close()


def test_device_job_builder_with_more_than_65535_chains(hip, oracle):
    """A read whose one SMEM has 66 000 occurrences 460 bp apart (no two of them chain together: the reference windows differ by
    more than w), with -c / max_occ raised above that: 66 000 chains of one seed each pass mem_chain_flt, more than the 16-bit
    index of the wave sort's rank keys holds, so the closing pass of the sort runs in its sequential form (chain_core.h).  The
    batch must still equal the host builder's, byte for byte, and the regions the oracle's."""
    import ctypes as C, torch
    from bwamem_hip import fmindex as F, synth, pipeline as P
    from bwamem_hip.lib import ChainOpt, ChainWorkspace, HostJobs, dev_jobs_to_host, seeds_to_host, load_library
    B = hip
    rng = np.random.default_rng(65)
    unit = rng.integers(0, 4, size=60).astype(np.uint8)
    n_copies, period = 66_000, 460
    g = rng.integers(0, 4, size=n_copies * period + 1000).astype(np.uint8)
    for k in range(n_copies):
        g[500 + k * period: 500 + k * period + 60] = unit
        g[500 + k * period - 1] = 0; g[500 + k * period + 60] = 1      # every copy sits between an A and a C ...
    n = len(g)
    dev = torch.device("cuda", 0)
    pac_t = F.pack_pac_device(torch.from_numpy(g).to(dev))
    d = F.build_fmd_index_device(pac_t, n, sa_intv=1, verify=True)
    dindex = B.Index.from_device(d.primary, d.L2.astype(np.uint64), d.seq_len, d.bwt_t, 1, d.sa_t, d.bits_t, pac_t=pac_t, l_pac=n)
    reads = rng.integers(0, 4, size=(4, 150)).astype(np.uint8)
    reads[1, 40:100] = unit                                   # the seed-rich read ...
    reads[1, 39] = 3; reads[1, 100] = 2                       # ... between a T and a G: no copy extends the match, the SMEM is the unit itself
    reads[2] = g[700_000:700_150]                             # ordinary reads around it
    reads[3] = synth.revcomp(g[900_000:900_150])
    flat, offs, lens = common.flat_reads(reads)
    dr = P.reads_to_device(reads, dev)
    ws = B.SeedWorkspace(4, 600, max_cands=600, max_occ=1 << 20)
    s = ws.seed_batch(dindex, dr.ascii, dr.offs, dr.lens, 19)
    sh = seeds_to_host(s, 4)
    assert int(sh["n_ref_pos"][1]) >= n_copies
    opt = ChainOpt(); load_library().bmh_chain_opt_default(C.byref(opt))
    opt.max_occ = 1 << 20
    cw = ChainWorkspace(4, int(s.n_seeds) + 64, opt=opt)
    dj = cw.chain_batch(dindex, dr.ascii, dr.offs, dr.lens, s)
    got = dev_jobs_to_host(dj, 4)
    hj = HostJobs(g, flat, offs, lens, sh, n_threads=4, opt=opt)
    assert int(dj.n_regs) == hj.n_regs and hj.n_regs > 65536 and int(dj.n_jobs) == hj.n_jobs
    for k in ("qlen", "tlen", "h0", "job_read", "job_reg", "job_side", "regs_per_read", "q", "t"):
        assert np.array_equal(got[k], getattr(hj, k)), k
    out3 = torch.zeros(hj.n_jobs + 1, 3, dtype=torch.int32, device=dev)
    regs = torch.zeros(hj.n_regs + 1, 8, dtype=torch.int32, device=dev)
    cw.extend(out3); cw.merge(out3, regs)
    torch.cuda.synchronize()
    want3, _, _ = oracle.extend_batch(*hj.jobs(), n_threads=4)
    assert np.array_equal(out3.cpu().numpy()[: hj.n_jobs], want3)
    assert np.array_equal(regs.cpu().numpy()[: hj.n_regs], hj.merge(want3))
    hj.free(); cw.free(); ws.free(); dindex.free()


def test_region_tail_on_the_device_equals_the_host_form(hip, oracle):
    """bmh_finalize_regs_device (lane per read; wave per read with the regions in LDS; in HBM beyond 512 regions) vs bmh_finalize_regs
    on the regions of the device pipeline: plain reads, a repeat-rich genome (reads with hundreds of regions), reads with a long
    deletion (two colinear regions merged through the patch test's global alignment), -a / another -T, three sequences, and a read
    with more regions than the wave kernel keeps in LDS."""
    import ctypes as C, torch
    from bwamem_hip import fmindex, synth
    from bwamem_hip.lib import ChainOpt, ChainWorkspace, PostOpt, dev_jobs_to_host, finalize_regs_device, load_library, _np_ptr, _u8p, _u64p, _i32p, _u32p
    B = hip
    Lb = load_library()
    rng = np.random.default_rng(12)

    def run(g, reads, contigs=None, po_over=None, co_over=None, want_wave=0, want_big=0):
        idx = fmindex.build_fmd_index(g)
        n, L = reads.shape
        flat = np.ascontiguousarray(reads.reshape(-1))
        pac = _pack_pac(g)
        dindex = B.Index.upload(idx, pac=pac, l_pac=len(g))
        ws = B.SeedWorkspace(n, n * L, max_cands=n * L, max_occ=1 << 22)
        r = _to_dev(torch, synth.codes_to_ascii(flat))
        o = (torch.arange(n, dtype=torch.int64) * L).to(torch.int32).cuda()
        l = torch.full((n,), L, dtype=torch.int32).cuda()
        s = ws.seed_batch(dindex, r, o, l, 19)
        co = ChainOpt(); Lb.bmh_chain_opt_default(C.byref(co))
        for k, v in (co_over or {}).items():
            setattr(co, k, v)
        cw = ChainWorkspace(n, max(int(s.n_seeds), 1), opt=co); cw.set_materialize(False)
        if contigs:
            cw.set_contigs(contigs)
        dj = cw.chain_batch(dindex, r, o, l, s)
        nr = int(dj.n_regs)
        regs = torch.zeros(nr + 1, 8, dtype=torch.int32, device="cuda")
        dj = cw.extend_merge(dindex, r, o, l, s, regs)
        torch.cuda.synchronize()
        dh = dev_jobs_to_host(dj, n)
        assert (dh["regs_per_read"] > 8).sum() >= want_wave and (dh["regs_per_read"] > 512).sum() >= want_big
        ep = B.ExtParams.default()
        po = PostOpt(); Lb.bmh_post_opt_default(C.byref(po)); po.id0 = 1234567
        for k, v in (po_over or {}).items():
            setattr(po, k, v)
        ctg = contigs or [("chrS", len(g))]
        c_off = np.ascontiguousarray(np.concatenate([[0], np.cumsum([c[1] for c in ctg])[:-1]]), dtype=np.int64)
        out = np.zeros((max(nr, 1), 16), np.int32); opr = np.zeros(n, np.uint32)
        fr = np.ascontiguousarray(dh["frac_rep"], dtype=np.float32)
        m = Lb.bmh_finalize_regs(C.byref(co), C.byref(ep), C.byref(po), len(g), _np_ptr(pac, _u8p), n, _np_ptr(flat, _u8p),
                                 _np_ptr(np.arange(n, dtype=np.uint64) * L, _u64p), _np_ptr(np.ascontiguousarray(regs.cpu().numpy()[:nr]), _i32p),
                                 _np_ptr(np.ascontiguousarray(dh["regs_per_read"]), _u32p), fr.ctypes.data_as(C.POINTER(C.c_float)),
                                 len(ctg), c_off.ctypes.data_as(C.c_void_p), _np_ptr(out, _i32p), _np_ptr(opr, _u32p), 4)
        assert m >= 0
        d_out, d_opr = finalize_regs_device(dindex, co, ep, po, r, o, regs, nr, dj.d_regs_per_read, dj.d_frac_rep, n, contigs=contigs)
        assert np.array_equal(d_opr.cpu().numpy().view(np.uint32)[:n], opr)
        got = d_out.cpu().numpy()
        assert got.shape == (m, 16) and np.array_equal(got, out[:m]), np.nonzero((got != out[:m]).any(1))[0][:5]
        cw.free(); ws.free(); dindex.free()
        return m, nr

    g = synth.make_genome(1_200_000, seed=3)
    reads, _ = synth.make_reads(g, 3000, 150, seed=9, sub_rate=0.02, indel_frac=0.3)
    for i in range(0, 500):                                  # a long deletion in the middle: two colinear regions -> the patch test
        p0 = int(rng.integers(0, len(g) - 400)); d = int(rng.integers(12, 60))
        x = np.concatenate([g[p0:p0 + 75], g[p0 + 75 + d:p0 + 150 + d]])
        reads[i] = x if i & 1 else synth.revcomp(x)
    m, nr = run(g, reads)
    assert m < nr - 100                                      # regions were merged
    run(g, reads, po_over=dict(flag_all=1, T=20))
    cuts = [0, 300_000, 700_000, len(g)]
    run(g, reads, contigs=[("c%d" % i, cuts[i + 1] - cuts[i]) for i in range(3)])
    gr = synth.make_genome(600_000, seed=9, repeat_frac=0.6, repeat_len=(200, 800), repeat_copies=(50, 400), repeat_div=0.02)
    readsr, _ = synth.make_reads(gr, 2000, 150, seed=8, sub_rate=0.01)
    run(gr, readsr, want_wave=50)
    run(gr, readsr, po_over=dict(flag_all=1), co_over=dict(max_occ=50, mask_level=0.3), want_wave=50)
    # 900 copies of a 60 bp unit, 460 bp apart, between an A and a C; the read carries it between a T and a G: one SMEM, 900 chains
    unit = rng.integers(0, 4, size=60).astype(np.uint8)
    gb = rng.integers(0, 4, size=900 * 460 + 1000).astype(np.uint8)
    for k in range(900):
        gb[500 + k * 460: 560 + k * 460] = unit; gb[499 + k * 460] = 0; gb[560 + k * 460] = 1
    rb = rng.integers(0, 4, size=(64, 150)).astype(np.uint8)
    for i in range(0, 64, 7):
        rb[i, 40:100] = unit; rb[i, 39] = 3; rb[i, 100] = 2
        rb[i, 50 + i % 40] = (rb[i, 50 + i % 40] + 1) & 3 if i % 14 == 0 else rb[i, 50 + i % 40]
    run(gb, rb, co_over=dict(max_occ=2000), want_big=3)


def test_mate_rescue_alignments_on_the_device(hip):
    """The ksw_align2 emulation of the mate rescue (csrc/pair_kernels.hip: 16 GPU lanes = the 16 byte lanes / 8 word lanes of the
    reference's striped SSE2 kernel) against the host walk of the same kernel (local_sw.cpp, itself checked against the compiled
    ksw_align2): score, ends, second-best score and row, start positions, for both widths, both strands, windows with and without
    the mate, repeats inside the window (second-best hits), N bases."""
    import ctypes as C, torch
    from bwamem_hip import fmindex, synth
    B = hip
    L = B.load_library()
    rng = np.random.default_rng(77)
    g = synth.make_genome(400_000, seed=5, repeat_frac=0.3, repeat_len=(100, 400), repeat_copies=(5, 40), repeat_div=0.03)
    idx = fmindex.build_fmd_index(g)
    dindex = B.Index.upload(idx, pac=_pack_pac(g), l_pac=len(g))
    n = len(g)
    text = np.concatenate([g, 3 - g[::-1]])                      # the 2 l_pac text the windows are cut from

    class Job(C.Structure):
        _fields_ = [("rb", C.c_int64), ("re", C.c_int64), ("read", C.c_uint32), ("l_ms", C.c_int32), ("is_rev", C.c_int32), ("xtra", C.c_int32), ("bl_off", C.c_uint32), ("pad", C.c_uint32)]
    XBYTE, XSUBO, XSTART = 0x10000, 0x40000, 0x80000
    for rl in (150, 100, 249, 300, 37):
        n_reads = 2400 if rl == 150 else 600
        reads = np.zeros((n_reads, rl), np.uint8)
        jobs = (Job * n_reads)()
        for i in range(n_reads):
            p0 = int(rng.integers(1000, n - rl - 1000))
            x = g[p0:p0 + rl].copy()
            m = rng.random(rl) < (0.02 if i % 3 else 0.12); x[m] = (x[m] + rng.integers(1, 4, size=int(m.sum()))) & 3
            if i % 11 == 0:
                x[int(rng.integers(0, rl))] = 4
            if i % 7 == 0 and rl >= 60:                             # a deletion in the mate
                k = int(rng.integers(20, rl - 20)); x = np.concatenate([x[:k], g[p0 + k + 3:p0 + rl + 3]])[:rl]
            if i % 4 == 1 and rl >= 60:                             # an insertion into the mate, anywhere -- also across the columns of two SSE lanes: F that flows from lane to lane
                k = int(rng.integers(5, rl - 5)); m_ = int(rng.integers(1, 5))
                x = np.concatenate([x[:k], rng.integers(0, 4, size=m_).astype(np.uint8), x[k:]])[:rl]
            is_rev = int(rng.integers(0, 2))
            reads[i] = synth.revcomp(x) if is_rev else x         # the job aligns the read's reverse complement when is_rev
            w0 = p0 - int(rng.integers(0, 400)) if i % 5 else int(rng.integers(0, n - 1000))       # every 5th window misses the mate
            w1 = w0 + int(rng.integers(rl // 2, 700))
            if i % 13 == 0:                                         # a window on the reverse strand of the text: the mate's complement lies there
                w0, w1 = 2 * n - w1, 2 * n - w0
                reads[i] = x if is_rev else synth.revcomp(x)
            jobs[i].rb, jobs[i].re, jobs[i].read, jobs[i].l_ms, jobs[i].is_rev = max(w0, 0), min(w1, 2 * n), i, rl, is_rev
            jobs[i].xtra = XSUBO | XSTART | (XBYTE if rl < 250 else 0) | 19
        r = _to_dev(torch, synth.codes_to_ascii(reads.reshape(-1)))
        o = (torch.arange(n_reads, dtype=torch.int64) * rl).to(torch.int32).cuda()
        L.bmh_matesw_batch_device.restype = C.c_int
        L.bmh_matesw_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        L.bmh_local_sw_c.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        # the reference's scoring, and (mates short enough for byte mode at a = 2) others: gap opens that differ, a free insertion open (the register form of
        # the kernel steps aside for that one: its F is the plain recurrence only where an open costs something)
        scorings = [None] + ([(2, 3, 5, 2, 5, 2), (1, 4, 6, 1, 2, 2), (1, 4, 6, 1, 0, 1)] if rl <= 100 else [])
        for sc in scorings:
            ep = B.ExtParams.default()
            if sc:
                ep.a, ep.b, ep.o_del, ep.e_del, ep.o_ins, ep.e_ins = sc
            want_all = np.zeros((n_reads, 7), np.int32)
            n_hit = n_sub = 0
            for i in range(n_reads):
                q = reads[i].copy()
                if jobs[i].is_rev:
                    q = synth.revcomp(q)
                t = np.ascontiguousarray(text[jobs[i].rb:jobs[i].re]); q = np.ascontiguousarray(q)
                L.bmh_local_sw_c(rl, q.ctypes.data_as(C.c_void_p), len(t), t.ctypes.data_as(C.c_void_p), C.byref(ep), jobs[i].xtra, want_all[i].ctypes.data_as(C.c_void_p))
                n_hit += want_all[i, 0] >= 19 and want_all[i, 6] >= 0; n_sub += want_all[i, 3] > 0
            assert n_hit > 200 and (rl < 100 or n_sub > 3), (rl, n_hit, n_sub)
            # both forms of the kernel: the lane's columns in registers (byte mode up to 256 columns, MSW_REG 1: the default) and in LDS
            for knob in (1, 0):
                L.bmh_tune_set(b"MSW_REG", knob, 0)
                out = np.zeros((n_reads, 7), np.int32)
                rc = L.bmh_matesw_batch_device(dindex.handle, r.data_ptr(), o.data_ptr(), C.byref(ep), C.byref(jobs), n_reads, out.ctypes.data_as(C.c_void_p), None)
                L.bmh_tune_set(b"MSW_REG", 0, 1)
                assert rc == 0, L.bmh_last_error()
                bad = np.nonzero((out != want_all).any(1))[0]
                assert bad.size == 0, (rl, sc, knob, bad[:5], out[bad[:3]], want_all[bad[:3]])
    dindex.free()


def test_mate_rescue_kernels_agree_on_hard_windows(hip):
    """The two forms of the mate rescue's kernel (columns in registers / in LDS, csrc/pair_kernels.hip) on 60 000 windows each of two shapes with diverged mates,
    insertions and deletions, N, windows without the mate and repeats (scripts/msw_bench.py with MSW_HARD: it asserts that every result word is equal and, where
    they are not, says which form left the host walk of the striped kernel).  This run found what the golden sets had not: insertions whose F crosses from the
    first SSE lane into the second (2 of 200 000 windows) were lost by the register form while a select sat behind its DPP shift."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for shape in (("60000", "150", "480"), ("40000", "249", "700")):
        r = subprocess.run([sys.executable, os.path.join(root, "scripts", "msw_bench.py"), *shape], env=dict(os.environ, MSW_HARD="1"), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "identical results: True" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])


def test_aligner_takes_registered_host_memory_and_times_its_copies(hip):
    """bmh_aligner_run with the letters of the read set in pageable memory (staged into the lanes' pinned buffers by host threads) and in REGISTERED host
    memory (bmh_host_pin = hipHostRegister: the batches go to the device straight from the caller's buffer): the same text; bmh_align_stats_t reports the
    copies themselves -- bytes in (letters + offsets + lengths + names) and out (the SAM text), seconds from events on the lanes' streams."""
    import ctypes as C
    from bwamem_hip import fmindex, synth
    from bwamem_hip.aligner import ReadSet
    from bwamem_hip.lib import NativeAligner, PeOpt, ChainOpt, PostOpt
    B = hip
    L = B.load_library()
    g = synth.make_genome(1_200_000, seed=23, repeat_frac=0.3)
    contigs = [("chrA", 700_000), ("chrB", 500_000)]
    dindex = B.Index.upload(fmindex.build_fmd_index(g), pac=_pack_pac(g), l_pac=len(g))
    dindex.densify_sa(1)
    n, rl = 6000, 150
    reads = synth.make_reads(g, n, rl, seed=3)[0]
    flat = np.ascontiguousarray(reads.reshape(-1))
    asc = synth.codes_to_ascii(flat)
    names = [("r%05d" % i) for i in range(n)]
    blob = np.frombuffer(("\0".join(names) + "\0").encode(), dtype=np.uint8)
    rs = ReadSet(asc, np.arange(n, dtype=np.uint64) * np.uint64(rl), np.full(n, rl, np.uint32), blob, np.arange(n, dtype=np.uint64) * np.uint64(7), codes=flat)
    co = ChainOpt(); L.bmh_chain_opt_default(C.byref(co)); po = PostOpt(); L.bmh_post_opt_default(C.byref(po)); pe_o = PeOpt(); L.bmh_pe_opt_default(C.byref(pe_o))
    nat = NativeAligner(dindex, _pack_pac(g), len(g), contigs, None, co, B.ExtParams.default(), po, pe_o)
    cuts = [0, 2000, 4000, n]
    L.bmh_host_pin.argtypes = [C.c_void_p, C.c_size_t]; L.bmh_host_unpin.argtypes = [C.c_void_p]
    texts, stats = [], []
    for pinned in (False, True):
        if pinned:
            assert L.bmh_host_pin(asc.ctypes.data, asc.nbytes) == 0, B.lib._err(L)
        try:
            parts = []
            st = nat.run(rs, cuts, False, lambda mv: parts.append(bytes(mv)), n_lanes=2, n_threads=4)
            texts.append(b"".join(parts)); stats.append(st)
        finally:
            if pinned:
                assert L.bmh_host_unpin(asc.ctypes.data) == 0
    assert texts[0] == texts[1] and texts[0].count(b"\n") >= n
    for st in stats:
        assert st.d2h_bytes == len(texts[0]) and st.h2d_bytes >= n * rl + 8 * n and st.h2d_copy_seconds > 0 and st.d2h_copy_seconds > 0
    nat.free(); dindex.free()
