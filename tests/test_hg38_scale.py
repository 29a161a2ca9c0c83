"""GPU tests of the device index builder (bmh_index_build) and of the hot path on a text beyond 2^32 symbols --
BASELINE.json configs[1]'s scale: the reference packs positions in 33 bits because hg38's fwd+revcomp text has 6.2e9 symbols
(/root/reference/src/GPUSeed/seed_gen.cu:943,1073-1077)."""
import os

import numpy as np
import pytest

import common

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hip():
    import torch
    import bwamem_hip as B
    B.load_library()
    assert torch.cuda.is_available(), "these tests need a GPU"
    return B


def _same(h, ref):
    return (h.primary == ref.primary and np.array_equal(h.L2, ref.L2) and np.array_equal(h.bwt_words, ref.bwt_words)
            and np.array_equal(h.sa, ref.sa) and np.array_equal(h.sa_bits, ref.sa_bits))


def test_device_builder_writes_the_reference_index_files(hip, tmp_path):
    """bmh_index_build on the genome of tests/golden/ref_index_g20011.fa -> .bwt / .sa byte-identical to the files the
    reference's own `bwa index` CLI wrote (both passes of build_index.sh; tests/golden/make_golden.py)."""
    import torch
    from bwamem_hip import fmindex as F, synth
    g = synth.make_genome(20011, seed=42)
    pac = F.pack_pac_device(torch.from_numpy(g).cuda())
    for cap in (None, "11"):                       # default: one pass; 2^11: 20 bucket passes, group-aligned doubling chunks
        if cap:
            os.environ["BMH_BUILD_CAP_LOG2"] = cap
        try:
            d = F.build_fmd_index_device(pac, len(g), sa_intv=16, verify=True)
        finally:
            os.environ.pop("BMH_BUILD_CAP_LOG2", None)
        p = str(tmp_path / "g")
        F.write_index(p, F.device_index_to_host(d, 16))
        for ext in (".bwt", ".sa"):
            assert open(p + ext, "rb").read() == open(os.path.join(common.GOLDEN, "ref_index_g20011" + ext), "rb").read(), (ext, cap)


def test_device_builder_matches_host_builder_on_hard_texts(hip):
    """repeat-rich, homopolymer and tandem texts (deep doubling: the depth reaches the text length), every sampling interval,
    small chunk capacities"""
    import torch
    from bwamem_hip import fmindex as F, synth
    rng = np.random.default_rng(1)
    cases = {"tiny": rng.integers(0, 4, 37, dtype=np.uint8), "one": np.array([2], np.uint8),
             "rep": synth.make_genome(100_000, seed=5, repeat_frac=0.5, repeat_div=0.01), "polyA": np.zeros(3000, np.uint8),
             "tandem": np.tile(rng.integers(0, 4, 171, dtype=np.uint8), 40), "palindrome": np.tile(np.array([0, 3], np.uint8), 500),
             "m1": synth.make_genome(1_000_000, seed=7, repeat_frac=0.3, repeat_div=0.02)}
    for name, g in cases.items():
        ref = F.build_fmd_index(g, sa_intv=16, device="cuda:0")
        pac = F.pack_pac_device(torch.from_numpy(g).cuda())
        for cap in ((None, "16") if name == "m1" else (None, "14") if name in ("polyA", "tandem", "palindrome") else (None, "12", "9")):
            if cap:
                os.environ["BMH_BUILD_CAP_LOG2"] = cap
            try:
                for intv in (16, 4, 1):
                    d = F.build_fmd_index_device(pac, len(g), sa_intv=intv, verify=True)
                    assert d.stats["verified"] == 1
                    assert _same(F.device_index_to_host(d, 16), ref), (name, cap, intv)
            finally:
                os.environ.pop("BMH_BUILD_CAP_LOG2", None)


def test_builder_reports_a_group_beyond_the_chunk_capacity(hip):
    import torch
    from bwamem_hip import fmindex as F
    pac = F.pack_pac_device(torch.zeros(5000, dtype=torch.uint8).cuda())
    os.environ["BMH_BUILD_CAP_LOG2"] = "9"
    try:
        with pytest.raises(RuntimeError, match="capacity"):
            F.build_fmd_index_device(pac, 5000, sa_intv=16)
    finally:
        os.environ.pop("BMH_BUILD_CAP_LOG2", None)


def test_hot_path_on_a_text_beyond_2_pow_32(hip, oracle):
    """A 2.2 Gbp hg38-like genome (24 contigs, 50 % repeats, N-runs): seq_len = 4.4e9 > 2^32.  Index built and completely
    verified on the device; seeds, extension jobs, extension results and regions of 3000 reads bit-identical to the oracle
    (which reads a host copy of the same index); positions beyond 2^32 really occur; reads are found where they were drawn."""
    import torch
    B = hip
    from bwamem_hip import fmindex as F, synth, pipeline as P
    from bwamem_hip.lib import ChainWorkspace, HostJobs, seeds_to_host
    dev = torch.device("cuda", 0)
    n = 2_200_000_000
    g_t, meta = synth.make_genome_device(n, dev, seed=11, return_meta=True)
    pac_t = F.pack_pac_device(g_t)
    g = g_t.cpu().numpy()
    del g_t
    torch.cuda.empty_cache()
    d = F.build_fmd_index_device(pac_t, n, sa_intv=1, verify=True)
    assert d.seq_len == 2 * n > 1 << 32 and d.stats["verified"] == 1
    dindex = B.Index.from_device(d.primary, d.L2.astype(np.uint64), d.seq_len, d.bwt_t, 1, d.sa_t, d.bits_t, pac_t=pac_t, l_pac=n)
    n_reads = 3000
    reads, truth = synth.make_reads(g, n_reads, 150, seed=7, holes=meta["holes"])
    # half of the reads from the first 100 Mbp: the reverse-strand text positions 2n - p of those lie beyond 2^32
    reads2, truth2 = synth.make_reads(g[meta["holes"][0][1]:100_000_000], n_reads // 2, 150, seed=8)
    reads[n_reads // 2:] = reads2
    truth["pos"][n_reads // 2:] = truth2["pos"] + meta["holes"][0][1]
    flat, offs, lens = common.flat_reads(reads)
    hidx = F.device_index_to_host(d, 16)
    want = oracle.seed_reads(oracle.fmd(hidx), flat, offs, lens, 19, n_threads=8)
    dr = P.reads_to_device(reads, dev)
    ws = B.SeedWorkspace(n_reads, n_reads * 150, max_cands=n_reads * 150)
    s = ws.seed_batch(dindex, dr.ascii, dr.offs, dr.lens, 19)
    common.assert_seeds_equal(seeds_to_host(s, n_reads), want, "beyond 2^32: ")
    assert int(want["rbeg"].max()) > 1 << 32
    # the same from the sparser samples of the reference's files (33rd bit of a sample through the packed bit array + LF walk)
    d16 = B.Index.upload(hidx, pac=None)
    s16 = ws.seed_batch(d16, dr.ascii, dr.offs, dr.lens, 19)
    common.assert_seeds_equal(seeds_to_host(s16, n_reads), want, "beyond 2^32, sa_intv 16: ")
    d16.densify_sa(4)
    s4 = ws.seed_batch(d16, dr.ascii, dr.offs, dr.lens, 19)
    common.assert_seeds_equal(seeds_to_host(s4, n_reads), want, "beyond 2^32, densified to 4: ")
    d16.free()
    s = ws.seed_batch(dindex, dr.ascii, dr.offs, dr.lens, 19)
    cw = ChainWorkspace(n_reads, int(s.n_seeds) + 64)
    cw.set_contigs(meta["contigs"])
    cw.set_materialize(False)
    dj = cw.chain_batch(dindex, dr.ascii, dr.offs, dr.lens, s)
    hj = HostJobs(g, flat, offs, lens, want, n_threads=8, contigs=meta["contigs"])
    assert int(dj.n_jobs) == hj.n_jobs and int(dj.n_regs) == hj.n_regs
    o3 = torch.zeros(hj.n_jobs + 1, 3, dtype=torch.int32, device=dev)
    r8 = torch.zeros(hj.n_regs + 1, 8, dtype=torch.int32, device=dev)
    cw.extend(o3)
    cw.merge(o3, r8)
    torch.cuda.synchronize()
    want3, _, _ = oracle.extend_batch(*hj.jobs(), n_threads=8)
    assert np.array_equal(o3.cpu().numpy()[: hj.n_jobs], want3)
    rg = r8.cpu().numpy()[: hj.n_regs]
    assert np.array_equal(rg, hj.merge(want3))
    r8b = torch.zeros(hj.n_regs + 1, 8, dtype=torch.int32, device=dev)
    cw.extend_merge(dindex, dr.ascii, dr.offs, dr.lens, s, r8b)                  # the one-call, two-pass form
    torch.cuda.synchronize()
    assert np.array_equal(r8b.cpu().numpy()[: hj.n_regs], rg)
    rb = rg[:, 4].view(np.uint32).astype(np.int64) | (rg[:, 5].astype(np.int64) << 32)
    re = rg[:, 6].view(np.uint32).astype(np.int64) | (rg[:, 7].astype(np.int64) << 32)
    assert (re > 1 << 32).mean() > 0.08
    fb = np.where(rb >= n, 2 * n - re, rb)
    fe = np.where(rb >= n, 2 * n - rb, re)
    tp = truth["pos"][rg[:, 0]]
    found = np.zeros(n_reads, bool)
    found[rg[:, 0][(fb < tp + 150) & (fe > tp)]] = True
    assert found.mean() > 0.99, found.mean()
    hj.free(); cw.free(); ws.free(); dindex.free()


@pytest.mark.parametrize("variant", ["single", "paired", "300bp"])
def test_baseline_configs_1_3_and_4_at_full_size(variant):
    """BASELINE.json configs[1] (1 M x 150 bp single-end: the headline workload), configs[3] (1 M x 150 bp pairs, interleaved) and configs[4] (1 M x 300 bp) at their stated size against the
    3.1 Gbp index (seq_len 6.2e9 > 2^32, built on the device in the run's setup): the bench command itself, two timed steps, the
    first 100 000 reads of the last timed batch compared with the oracle inside the run (seeds and regions identical or exit 3)."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    # (configs[3] is 1 M PAIRS: two million interleaved reads per batch)
    extra = {"single": [], "paired": ["--paired", "--reads-per-gpu", "2000000"], "300bp": ["--read-len", "300"]}[variant] + ["--distinct-batches", "2"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--verify-sample", "100000", "--no-next-rows", "--cpu-sample", "0",
                        "--no-pcie"] + extra, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    res = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][-1])
    cfg = res["config"]
    assert cfg["reads_per_gpu"] == (2_000_000 if variant == "paired" else 1_000_000) and cfg["seq_len"] > 2**32 and cfg["genome_mbp"] == 3100
    assert cfg["read_len"] == (300 if variant == "300bp" else 150) and cfg["paired_interleaved"] == (variant == "paired")
    v = res["verified"]
    assert v["reads"] == 100_000 and v["seeds_identical"] and v["regions_identical"], v
    assert res["value"] > 0


@pytest.mark.parametrize("shape", ["pairs_150bp", "single_300bp"])
def test_reference_host_code_at_hg38_scale(hip, tmp_path, shape):
    """The REFERENCE's own host code against the device-resident path on the 3.1 Gbp genome (24 sequences, seq_len 6.2e9 > 2^32; index built by
    bmh_index_build and written in the reference's file layout for both sides to load): build/dropin/bwa-gasal2 -- the reference's src/*.c compiled
    unchanged against include/seed_gen.h + include/gasal2_root, linked with libbwamem_hip.so -- at -t 1 beside bwamem_hip.aligner on 100 000 hard
    pairs of 150 bp (configs[3]'s shape: diverged / relocated / random / chimeric mates) and on 50 000 hard single-end reads of 300 bp (configs[4]'s):
    every SAM record identical.  This is the comparison of the host-side record logic (positions, TLEN, pairing windows, the mate rescue's windows,
    CIGARs) with the reference's own code at coordinates that do not fit 32 bits (scripts/e2e_hg.sh baseline, now inside the driver-run suite)."""
    import subprocess
    import sys
    if not os.path.exists(os.path.join(ROOT, "build", "dropin", "bwa-gasal2")):
        pytest.skip("build/dropin/bwa-gasal2 not built (needs /root/reference at build time)")
    n, mode, rl = ("200000", "pe_hard", "150") if shape == "pairs_150bp" else ("50000", "se_hard", "300")
    env = dict(os.environ, E2E_CONTIGS="24", E2E_NATIVE_BUILD="1", E2E_READLEN=rl, E2E_TAG="gputest_" + shape,
               E2E_GENOME_KW="{'repeat_frac': 0.3, 'repeat_copies': (10, 3000), 'repeat_len': (300, 3000), 'repeat_div': 0.03}")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "e2e_dropin.py"), str(tmp_path), "3100000000", n, "1", mode],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=1500)
    out = r.stdout.decode()
    assert r.returncode == 0 and "E2E DROP-IN OK" in out and "SAM IDENTICAL" in out, out[-3000:]
    assert "differing records: 0" in out, out[-3000:]


def test_reads_to_sam_device_forms_equal_host_forms_beyond_2_pow_32(hip):
    """bmh_aligner_run on a 2.2 Gbp hg38-like genome (24 sequences, seq_len 4.4e9, 50 % repeats), 200 000 single-end reads and 100 000 pairs: the
    text the device writes (records selected, CIGARs packed, SAM assembled on the device; pairs: mem_pair / mem_sam_pe's choices on the device for
    the pairs the rescue leaves alone) is byte for byte the text of the host forms (bmh_sam_need_cigar + bmh_format_sam[_pe]; all pairs through the
    host walks of mem_sam_pe), at positions beyond 2^32 and on reads with hundreds of hits.  Then the same with the last three sequences flagged as
    ALT contigs: the device tail's second marking round and pair_kernel's ALT rules against the host tail (BMH_ALIGNER_ALT_HOST_PATCH) and the host's
    mem_sam_pe (BMH_ALIGNER_PE_HOST)."""
    import ctypes as C
    import torch
    B = hip
    from bwamem_hip import fmindex as F, synth
    from bwamem_hip.aligner import ReadSet
    from bwamem_hip.lib import NativeAligner, PeOpt, ChainOpt, PostOpt
    dev = torch.device("cuda", 0)
    L = B.load_library()
    n = 2_200_000_000
    g_t, meta = synth.make_genome_device(n, dev, seed=11, return_meta=True)
    pac_t = F.pack_pac_device(g_t)
    g = g_t.cpu().numpy()
    del g_t
    torch.cuda.empty_cache()
    d = F.build_fmd_index_device(pac_t, n, sa_intv=1)
    dindex = B.Index.from_device(d.primary, d.L2.astype(np.uint64), d.seq_len, d.bwt_t, 1, d.sa_t, d.bits_t, pac_t=pac_t, l_pac=n)
    co = ChainOpt(); L.bmh_chain_opt_default(C.byref(co)); po = PostOpt(); L.bmh_post_opt_default(C.byref(po))
    pe_o = PeOpt(); L.bmh_pe_opt_default(C.byref(pe_o))
    pac_h = pac_t.cpu().numpy()
    nat = NativeAligner(dindex, pac_h, n, meta["contigs"], None, co, B.ExtParams.default(), po, pe_o)
    # the same index with its last three sequences flagged as ALT contigs: the device tail's and the pairing kernel's ALT rules against the host's
    is_alt = np.zeros(len(meta["contigs"]), np.uint8); is_alt[-3:] = 1
    nat_alt = NativeAligner(dindex, pac_h, n, meta["contigs"], is_alt, co, B.ExtParams.default(), po, pe_o)
    nth = L.bmh_effective_cpus()
    n_reads, rl = 200_000, 150
    for paired, alt in ((False, False), (True, False), (False, True), (True, True)):
        reads = (synth.make_pairs(g, n_reads // 2, rl, seed=21, holes=meta["holes"]) if paired else synth.make_reads(g, n_reads, rl, seed=21, holes=meta["holes"]))[0]
        flat = np.ascontiguousarray(np.asarray(reads, np.uint8).reshape(-1))
        w = len(str(n_reads))
        names = np.char.add("r", np.char.zfill((np.arange(n_reads) // (2 if paired else 1)).astype(str), w))
        blob = np.frombuffer(("\0".join(names.tolist()) + "\0").encode(), dtype=np.uint8)
        rs = ReadSet(synth.codes_to_ascii(flat), np.arange(n_reads, dtype=np.uint64) * np.uint64(rl), np.full(n_reads, rl, np.uint32), blob,
                     np.arange(n_reads, dtype=np.uint64) * np.uint64(w + 2), codes=flat)
        cuts = [0, (n_reads // 3) & ~1, (2 * n_reads // 3) & ~1, n_reads]
        texts = {}
        for env in ("", "BMH_ALIGNER_HOST_FORMAT") + (("BMH_ALIGNER_PE_HOST",) if paired else ()) + (("BMH_ALIGNER_ALT_HOST_PATCH",) if alt and not paired else ()):
            if env:
                os.environ[env] = "1"
            try:
                parts = []
                (nat_alt if alt else nat).run(rs, cuts, paired, lambda mv: parts.append(bytes(mv)), n_lanes=2, n_threads=nth)
                texts[env] = b"".join(parts)
            finally:
                if env:
                    del os.environ[env]
        body = texts[""]
        assert body.count(b"\n") >= n_reads and b"\tXA:Z:" in body
        if alt:
            assert body.count(b"\tpa:f:") > 200, body.count(b"\tpa:f:")
        pos = np.array([int(l.split(b"\t")[3]) for l in body.split(b"\n")[:20000] if l and l.split(b"\t")[2] != b"*"])
        assert pos.max() > 100_000_000
        for env, t in texts.items():
            if t != body:
                a, b = body.split(b"\n"), t.split(b"\n")
                assert False, (paired, alt, env, len(a), len(b), [(x, y) for x, y in zip(a, b) if x != y][:2])
    nat.free(); nat_alt.free(); dindex.free()
