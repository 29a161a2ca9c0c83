"""CPU test of the host job builder (bmh_build_jobs is host code: no GPU needed): it reproduces the exact
multiset of extension jobs that the REFERENCE's own host code submitted for the same reads and seeds
(tests/golden/jobs_golden.npz, recorded on the MI355X box by scripts/make_jobs_golden.py through
BMH_GASAL_DUMP), and with the oracle as extension back-end the best region score of every read equals the
AS tag of the reference's SAM output."""
import hashlib
import os

import numpy as np
import pytest

import common
from bwamem_hip import synth
from bwamem_hip.lib import HostJobs


def test_jobs_and_region_scores_match_reference_host_code(oracle):
    z = np.load(os.path.join(common.GOLDEN, "jobs_golden.npz"))
    g = synth.make_genome(int(z["n_genome"]), seed=int(z["genome_seed"]))
    reads = z["reads"]
    n, L = reads.shape
    seeds = {k: z[k] for k in ("rbeg", "qbeg", "score", "n_ref_pos", "prefix")}
    hj = HostJobs(g, reads.reshape(-1), np.arange(n, dtype=np.uint64) * L, np.full(n, L, np.uint32), seeds, n_threads=4)
    digs = []
    for i in range(hj.n_jobs):
        h0 = int(hj.h0[i])
        digs.append(hashlib.sha1(bytes([h0 & 255, h0 >> 8]) + hj.q[hj.qoff[i]:hj.qoff[i] + hj.qlen[i]].tobytes() + b"|" +
                                 hj.t[hj.toff[i]:hj.toff[i] + hj.tlen[i]].tobytes()).digest())
    digs.sort()
    want = [bytes(r) for r in z["job_digests"]]
    assert len(digs) == len(want) and digs == want
    # one thread or four: same batch
    hj1 = HostJobs(g, reads.reshape(-1), np.arange(n, dtype=np.uint64) * L, np.full(n, L, np.uint32), seeds, n_threads=1)
    assert hj1.n_jobs == hj.n_jobs and np.array_equal(hj1.q, hj.q) and np.array_equal(hj1.toff, hj.toff)
    # extension by the checker, region merge by the library: best score per read == AS of the reference's SAM
    out3, _, _ = oracle.extend_batch(*hj.jobs())
    regs = hj.merge(out3)
    best = np.full(n, -1, np.int64)
    np.maximum.at(best, regs[:, 0], regs[:, 1])
    as_tag = z["as_tag"]
    has = as_tag >= 0
    assert has.sum() > 0.9 * n and np.array_equal(best[has], as_tag[has])
    hj.free(); hj1.free()


def _chain_core_lib():
    """tests/chain_core_host.cpp: the per-read core of the DEVICE job builder (csrc/chain_core.h) compiled as plain C++."""
    import ctypes as C, subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(here, "_build"); os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "chain_core_host.so")
    src = [os.path.join(here, "chain_core_host.cpp"), os.path.join(here, "..", "bwa-mem_gpu_amd", "csrc", "chain_core.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I", os.path.join(here, "..", "include"), src[0], "-o", so])
    lib = C.CDLL(so)

    class Res(C.Structure):
        _fields_ = [(k, C.c_uint64) for k in ("n_regs", "n_jobs", "q_bytes", "t_bytes")] + \
                   [(k, C.POINTER(C.c_uint32)) for k in ("regs_per_read", "qoff", "qlen", "toff", "tlen", "h0", "job_read", "job_reg", "job_side")] + \
                   [("q", C.POINTER(C.c_uint8)), ("t", C.POINTER(C.c_uint8)), ("err", C.c_int)]
    lib.chain_core_run.restype = C.POINTER(Res)
    lib.chain_core_run_compact.restype = C.POINTER(Res)
    lib.chain_core_free.argtypes = [C.POINTER(Res)]
    return lib


def _run_core(lib, opt, g, reads, seeds, compact=False):
    import ctypes as C
    n, L = reads.shape
    pad = (-len(g)) % 4
    codes = np.concatenate([g, np.zeros(pad, np.uint8)]).reshape(-1, 4)
    pac = np.ascontiguousarray(((codes[:, 0] << 6) | (codes[:, 1] << 4) | (codes[:, 2] << 2) | codes[:, 3]).astype(np.uint8))
    a = lambda x, dt: np.ascontiguousarray(x, dtype=dt)
    keep = [pac, a(reads.reshape(-1), np.uint8), np.arange(n, dtype=np.uint64) * L, np.full(n, L, np.uint32), a(seeds["rbeg"], np.uint64),
            a(seeds["qbeg"], np.int32), a(seeds["score"], np.uint32), a(seeds["n_ref_pos"], np.uint32), a(seeds["prefix"], np.uint32)]
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    fn = lib.chain_core_run_compact if compact else lib.chain_core_run
    rp = fn(C.byref(opt), C.c_int64(len(g)), p(keep[0]), C.c_uint32(n), *[p(k) for k in keep[1:]], C.c_uint64(len(keep[4])))
    r = rp.contents
    arr = lambda ptr, nn: np.ctypeslib.as_array(ptr, shape=(max(int(nn), 1),))[:int(nn)].copy()
    out = {k: arr(getattr(r, k), r.n_jobs) for k in ("qoff", "qlen", "toff", "tlen", "h0", "job_read", "job_reg", "job_side")}
    out.update(q=arr(r.q, r.q_bytes), t=arr(r.t, r.t_bytes), regs_per_read=arr(r.regs_per_read, n), n_jobs=int(r.n_jobs), n_regs=int(r.n_regs), err=int(r.err))
    lib.chain_core_free(rp)
    return out


def test_device_chain_core_matches_reference_jobs_and_host_builder(oracle):
    """The chaining core the HIP kernels instantiate (csrc/chain_core.h), compiled for the host: (1) reproduces the
    golden job multiset of the reference's host code; (2) equals bmh_build_jobs array for array on a repeat-rich
    genome and with non-default options."""
    import ctypes as C
    from bwamem_hip import fmindex
    from bwamem_hip.lib import ChainOpt, load_library
    lib = _chain_core_lib()
    z = np.load(os.path.join(common.GOLDEN, "jobs_golden.npz"))
    g = synth.make_genome(int(z["n_genome"]), seed=int(z["genome_seed"]))
    opt = ChainOpt(); load_library().bmh_chain_opt_default(C.byref(opt))
    seeds = {k: z[k] for k in ("rbeg", "qbeg", "score", "n_ref_pos", "prefix")}
    c = _run_core(lib, opt, g, z["reads"], seeds)
    assert c["err"] == 0
    digs = sorted(hashlib.sha1(bytes([int(c["h0"][i]) & 255, int(c["h0"][i]) >> 8]) + c["q"][c["qoff"][i]:c["qoff"][i] + c["qlen"][i]].tobytes() + b"|" +
                               c["t"][c["toff"][i]:c["toff"][i] + c["tlen"][i]].tobytes()).digest() for i in range(c["n_jobs"]))
    assert digs == [bytes(r) for r in z["job_digests"]]
    # the same with the COMPACT scratch records of the cooperative kernels (round 6: 16-bit links and read coordinates, csrc/chain_core.h ch_compact_ty)
    cc = _run_core(lib, opt, g, z["reads"], seeds, compact=True)
    assert cc["err"] == 0 and all(np.array_equal(cc[k], c[k]) for k in ("qoff", "qlen", "toff", "tlen", "h0", "job_read", "job_reg", "job_side", "regs_per_read", "q", "t"))
    # repeat-rich genome, default and non-default options, against the host builder
    gr = synth.make_genome(400_000, seed=9, repeat_frac=0.6, repeat_len=(200, 800), repeat_copies=(50, 400), repeat_div=0.02)
    idx = fmindex.build_fmd_index(gr)
    reads, _ = synth.make_reads(gr, 1500, 150, seed=8, sub_rate=0.01)
    flat, offs, lens = common.flat_reads(reads)
    s = oracle.seed_reads(oracle.fmd(idx), flat, offs, lens, 19, n_threads=4)
    assert s["n_ref_pos"].max() > 100
    # (min_chain_weight 5 and 6: 1.1 W <= 0.05 * 150, so the reference's seed filter mem_flt_chained_seeds / mem_seed_sw runs, src/bwamem.c:970-991)
    for over in ({}, dict(max_occ=20), dict(max_occ=5, max_chain_extend=3, min_chain_weight=30, drop_ratio=0.9, mask_level=0.2),
                 dict(min_chain_weight=5), dict(min_chain_weight=6, max_occ=20, a=2, b=5, o_del=4, e_del=2, o_ins=7, e_ins=1)):
        o = ChainOpt(); load_library().bmh_chain_opt_default(C.byref(o))
        for k, v in over.items():
            setattr(o, k, v)
        c = _run_core(lib, o, gr, reads, s)
        hj = HostJobs(gr, flat, offs, lens, s, n_threads=4, opt=o)
        assert c["n_jobs"] == hj.n_jobs and c["n_regs"] == hj.n_regs
        for k in ("qoff", "qlen", "toff", "tlen", "h0", "job_read", "job_reg", "job_side", "regs_per_read", "q", "t"):
            assert np.array_equal(c[k], getattr(hj, k)), (over, k)
        if not over.get("min_chain_weight", 0) or over["min_chain_weight"] >= 30:      # (the compact records are for the forms without the seed filter)
            cc = _run_core(lib, o, gr, reads, s, compact=True)
            for k in ("qoff", "qlen", "toff", "tlen", "h0", "job_read", "job_reg", "job_side", "regs_per_read", "q", "t"):
                assert np.array_equal(cc[k], getattr(hj, k)), ("compact", over, k)
        hj.free()


def _golden_regions(oracle, z):
    """regions of the golden read set: host job builder on the stored seeds, oracle extension, host merge"""
    import ast
    g = synth.make_genome(int(z["n_genome"]), seed=int(z["genome_seed"]), **(ast.literal_eval(str(z["genome_kw"])) if "genome_kw" in z.files else {}))
    reads = z["reads"]; n, L = reads.shape
    seeds = {k: z[k] for k in ("rbeg", "qbeg", "score", "n_ref_pos", "prefix")}
    contigs = ast.literal_eval(str(z["contigs"])) if "contigs" in z.files else None
    hj = HostJobs(g, reads.reshape(-1), np.arange(n, dtype=np.uint64) * L, np.full(n, L, np.uint32), seeds, n_threads=4, contigs=contigs)
    out3, _, _ = oracle.extend_batch(*hj.jobs())
    return g, reads, hj, hj.merge(out3)


def _pac(g):
    pad = (-len(g)) % 4
    codes = np.concatenate([g, np.zeros(pad, np.uint8)]).reshape(-1, 4)
    return np.ascontiguousarray(((codes[:, 0] << 6) | (codes[:, 1] << 4) | (codes[:, 2] << 2) | codes[:, 3]).astype(np.uint8))


@pytest.mark.parametrize("golden", ["post_golden.npz", "contigs_golden.npz"])
def test_finalize_regs_matches_reference_sam(oracle, golden):
    """bmh_finalize_regs (mem_sort_dedup_patch, mem_mark_primary_se, mem_approx_mapq_se, the selection of mem_reg2sam)
    + the oracle's mem_reg2aln reproduce EVERY SAM record the reference's own host code wrote for a repeat-rich read
    set -- default run and -a run (secondary alignments): flag, RNAME, POS, MAPQ, CIGAR, NM, AS, XS, MD, in file order.
    contigs_golden: the same genome cut into three sequences, an eighth of the reads across a cut (the job builder drops
    seeds that bridge two sequences and clips extension windows to one, src/bwamem.c:437, src/bntseq.c:531-556)."""
    import ast
    z = np.load(os.path.join(common.GOLDEN, golden))
    g, reads, hj, regs = _golden_regions(oracle, z)
    pac = _pac(g)
    contigs = ast.literal_eval(str(z["contigs"])) or [("chrS", len(g))]
    c_off = np.concatenate([[0], np.cumsum([c[1] for c in contigs])])
    # the extension jobs themselves: same multiset as the reference submitted
    digs = sorted(hashlib.sha1(bytes([int(hj.h0[i]) & 255, int(hj.h0[i]) >> 8]) + hj.q[hj.qoff[i]:hj.qoff[i] + hj.qlen[i]].tobytes() + b"|" +
                               hj.t[hj.toff[i]:hj.toff[i] + hj.tlen[i]].tobytes()).digest() for i in range(hj.n_jobs))
    assert digs == [bytes(r) for r in z["job_digests"]]
    for tag, flag_all in (("def_", False), ("all_", True)):
        out, per_read = hj.finalize(regs, flag_all=flag_all, n_threads=2)
        assert per_read.sum() == len(out)
        got = []
        seen = set()
        for q in out:
            if not q[15]:
                continue
            rb = int(np.uint32(q[4])) | (int(q[5]) << 32); re = int(np.uint32(q[6])) | (int(q[7]) << 32)
            a = oracle.reg2aln(pac, len(g), reads[q[0]], q[2], q[3], rb, re, q[8], reg_w=int(q[9]))
            cs = "".join(f"{int(x) >> 4}{'MIDSH'[int(x) & 0xf]}" for x in a["cigar"])
            if int(q[0]) in seen:
                cs = cs.replace("S", "H")                       # every record of a read after its first is hard-clipped (mem_aln2sam)
            seen.add(int(q[0]))
            rid = int(np.searchsorted(c_off, a["pos"], side="right") - 1)
            got.append((int(q[0]), (16 if a["is_rev"] else 0) | int(q[14]), a["pos"] - int(c_off[rid]) + 1, int(q[13]), cs, a["NM"], int(q[1]),
                        int(q[10]) if q[12] < 0 else -1, a["MD"], contigs[rid][0]))
        for r in sorted(set(range(len(reads))) - seen):             # reads without a reported alignment: the unmapped record
            got.append((r, 4, 0, 0, "*", -1, 0, 0, "", "*"))                 # mem_aln2sam prints AS:i:0 XS:i:0 for it
        got.sort(key=lambda t: t[0])                                # (stable: the records of a read keep their order)
        rn = [str(x) for x in z[tag + "rname"]] if tag + "rname" in z.files else [contigs[0][0]] * len(z[tag + "read"])
        want = list(zip(z[tag + "read"].tolist(), z[tag + "flag"].tolist(), z[tag + "pos"].tolist(), z[tag + "mapq"].tolist(), [str(x) for x in z[tag + "cigar"]],
                        z[tag + "nm"].tolist(), z[tag + "as_"].tolist(), z[tag + "xs"].tolist(), [str(x) for x in z[tag + "md"]], rn))
        assert len(got) == len(want), (tag, len(got), len(want))
        bad = [(a, b) for a, b in zip(got, want) if a != b]
        assert not bad, (tag, len(bad), bad[:3])
    assert (z["all_flag"] & 0x100).sum() > 100 and (z["def_xs"] > 0).sum() > 100
    if golden.startswith("contigs"):
        assert len(set(str(x) for x in z["def_rname"])) == 4        # three sequences and '*'
    hj.free()


@pytest.mark.parametrize("golden", ["jobs_golden.npz", "post_golden.npz", "contigs_golden.npz"])
def test_sam_text_matches_reference(oracle, golden):
    """bmh_finalize_regs + bmh_sam_need_cigar + bmh_format_sam (CIGARs from the oracle here; from the device in the GPU
    suite) write the records of the reference's SAM file BYTE FOR BYTE: plain, repeat-rich (XS, XA tags, low MAPQ) and
    three-sequence golden read sets."""
    import ast, ctypes as C
    from bwamem_hip.lib import PostOpt, format_sam, load_library, _np_ptr, _i32p, _u32p, _u8p
    z = np.load(os.path.join(common.GOLDEN, golden))
    g, reads, hj, regs = _golden_regions(oracle, z)
    pac = _pac(g)
    contigs = (ast.literal_eval(str(z["contigs"])) if "contigs" in z.files else None) or [("chrS", len(g))]
    fin, per_read = hj.finalize(regs, flag_all=False, n_threads=2)
    L = load_library()
    po = PostOpt(); L.bmh_post_opt_default(C.byref(po))
    need = np.zeros(max(len(fin), 1), np.uint8)
    fin_c = np.ascontiguousarray(fin); pr_c = np.ascontiguousarray(per_read)
    m = L.bmh_sam_need_cigar(C.byref(po), _np_ptr(fin_c, _i32p), _np_ptr(pr_c, _u32p), len(per_read), _np_ptr(need, _u8p))
    idx = np.nonzero(need[: len(fin)])[0]
    assert m == len(idx)
    slot = np.full(max(len(fin), 1), -1, np.int64); slot[idx] = np.arange(len(idx))
    aln = np.zeros((max(len(idx), 1), 8), np.int32); cigar = np.zeros((max(len(idx), 1), 48), np.uint32); md = np.zeros((max(len(idx), 1), 640), np.uint8)
    for k, i in enumerate(idx):
        q = fin[i]
        rb = int(np.uint32(q[4])) | (int(q[5]) << 32); re = int(np.uint32(q[6])) | (int(q[7]) << 32)
        a = oracle.reg2aln(pac, len(g), reads[q[0]], q[2], q[3], rb, re, q[8], reg_w=int(q[9]))
        aln[k] = [a["pos"] & 0xFFFFFFFF if a["pos"] < 2 ** 31 else a["pos"] - 2 ** 32, a["pos"] >> 32, a["is_rev"], len(a["cigar"]), a["NM"], a["score"], len(a["MD"]), 0]
        cigar[k, : len(a["cigar"])] = a["cigar"]
        md[k, : len(a["MD"])] = np.frombuffer(a["MD"].encode(), np.uint8)
    n, rl = reads.shape
    txt = format_sam(po, [f"r{i}" for i in range(n)], reads.reshape(-1), np.arange(n, dtype=np.uint64) * rl, np.full(n, rl, np.uint32), contigs,
                     fin, per_read, slot, aln, cigar, md)
    want = bytes(z["sam_text"]).decode()
    if txt != want:
        gl, wl = txt.split("\n"), want.split("\n")
        bad = [(a, b) for a, b in zip(gl, wl) if a != b]
        assert False, (len(gl), len(wl), bad[:2])
    hj.free()


@pytest.mark.parametrize("golden", ["pe_golden.npz", "pe_contigs_golden.npz"])
def test_paired_end_sam_text_matches_reference(oracle, golden):
    """bmh_finalize_pairs (insert-size statistics, mate rescue with the host local alignment, pairing, mem_sam_pe's choices)
    + bmh_format_sam_pe write the reference's PAIRED-END SAM records (gase_aln -p on interleaved pairs) byte for byte:
    proper pairs, a mate found only by rescue, discordant and unmapped mates."""
    import ctypes as C
    from bwamem_hip.lib import ChainOpt, ExtParams, PostOpt, finalize_pairs, format_sam, load_library, _np_ptr, _i32p, _u32p, _u8p
    import ast
    z = np.load(os.path.join(common.GOLDEN, golden))
    g, reads, hj, regs = _golden_regions(oracle, z)
    pac = _pac(g)
    contigs = ast.literal_eval(str(z["contigs"])) or [("chrS", len(g))]
    n, rl = reads.shape
    L = load_library()
    co = ChainOpt(); L.bmh_chain_opt_default(C.byref(co)); ep = ExtParams.default(); po = PostOpt(); L.bmh_post_opt_default(C.byref(po))
    flat = np.ascontiguousarray(reads.reshape(-1)); offs = np.arange(n, dtype=np.uint64) * rl; lens = np.full(n, rl, np.uint32)
    fin, per_read, h_rec, unflag, pes = finalize_pairs(co, ep, po, len(g), pac, flat, offs, lens, regs, hj.regs_per_read, hj.frac_rep(),
                                                       contigs=contigs if len(contigs) > 1 else None, n_threads=2)
    assert pes[1][2] == 0 and 300 < pes[1][3] < 400          # FR orientation: mean insert ~350
    need = np.zeros(max(len(fin), 1), np.uint8)
    fin_c = np.ascontiguousarray(fin); pr_c = np.ascontiguousarray(per_read); h_c = np.ascontiguousarray(h_rec)
    m = L.bmh_sam_need_cigar_pe(C.byref(po), _np_ptr(fin_c, _i32p), _np_ptr(pr_c, _u32p), _np_ptr(h_c, _i32p), n, _np_ptr(need, _u8p))
    idx = np.nonzero(need[: len(fin)])[0]
    assert m == len(idx)
    slot = np.full(max(len(fin), 1), -1, np.int64); slot[idx] = np.arange(len(idx))
    aln = np.zeros((max(len(idx), 1), 8), np.int32); cigar = np.zeros((max(len(idx), 1), 48), np.uint32); md = np.zeros((max(len(idx), 1), 640), np.uint8)
    for k, i in enumerate(idx):
        q = fin[i]
        rb = int(np.uint32(q[4])) | (int(q[5]) << 32); re = int(np.uint32(q[6])) | (int(q[7]) << 32)
        a = oracle.reg2aln(pac, len(g), reads[q[0]], q[2], q[3], rb, re, q[8], reg_w=int(q[9]))
        aln[k] = [a["pos"], a["pos"] >> 32, a["is_rev"], len(a["cigar"]), a["NM"], a["score"], len(a["MD"]), 0]
        cigar[k, : len(a["cigar"])] = a["cigar"]
        md[k, : len(a["MD"])] = np.frombuffer(a["MD"].encode(), np.uint8)
    txt = format_sam(po, [f"p{i // 2}" for i in range(n)], flat, offs, lens, contigs, fin, per_read, slot, aln, cigar, md, h_rec=h_rec, unflag=unflag)
    want = bytes(z["sam_text"]).decode()
    if txt != want:
        gl, wl = txt.split("\n"), want.split("\n")
        bad = [(a, b) for a, b in zip(gl, wl) if a != b]
        assert False, (len(gl), len(wl), len(bad), bad[:3])
    hj.free()
