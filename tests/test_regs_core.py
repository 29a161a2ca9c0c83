"""CPU test of the region tail's device core (csrc/regs_core.h compiled as plain C++, tests/regs_core_host.cpp) against the host
form bmh_finalize_regs (regs_post.cpp, itself pinned to the reference's SAM records by the golden tests): same records, same order."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import common
from bwamem_hip import fmindex, synth
from bwamem_hip.lib import ChainOpt, ExtParams, HostJobs, PostOpt, load_library

HERE = os.path.dirname(os.path.abspath(__file__))


def core_lib():
    out = os.path.join(HERE, "_build"); os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "regs_core_host.so")
    src = [os.path.join(HERE, "regs_core_host.cpp"), os.path.join(HERE, "..", "bwa-mem_gpu_amd", "csrc", "regs_core.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I", os.path.join(HERE, "..", "include"), src[0], "-o", so])
    lib = C.CDLL(so)
    lib.regs_core_run.restype = C.c_int64
    return lib


def both(lib, L, co, ep, po, g, pac, flat, offs, regs, rpr, fr, contigs=None):
    n = len(rpr)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    ctg_off = np.ascontiguousarray(np.concatenate([[0], np.cumsum([c[1] for c in contigs])[:-1]]), dtype=np.int64) if contigs else None
    nct = len(contigs) if contigs else 1
    outs = []
    for fn, last in ((L.bmh_finalize_regs, 2), (lib.regs_core_run, 1)):
        out = np.full((max(len(regs), 1), 16), -3, np.int32); opr = np.zeros(max(n, 1), np.uint32)
        fn.restype = C.c_int64
        fn.argtypes = None
        m = fn(C.byref(co), C.byref(ep), C.byref(po), C.c_int64(len(g)), p(pac), C.c_uint32(n), p(flat), p(offs), p(regs), p(rpr), p(fr),
               C.c_int(nct), p(ctg_off) if contigs else None, p(out), p(opr), C.c_int(last))
        assert m >= 0, m
        outs.append((out[:m].copy(), opr[:n].copy()))
    return outs


def pack(g):
    pad = (-len(g)) % 4
    codes = np.concatenate([g, np.zeros(pad, np.uint8)]).reshape(-1, 4)
    return np.ascontiguousarray(np.concatenate([((codes[:, 0] << 6) | (codes[:, 1] << 4) | (codes[:, 2] << 2) | codes[:, 3]).astype(np.uint8), np.zeros(8, np.uint8)]))


@pytest.mark.parametrize("case", ["plain", "repeats", "indels_flag_all", "scoring"])
def test_region_tail_core_equals_host_form(oracle, case):
    L = load_library()
    lib = core_lib()
    kw, rkw, over, pover, scoring = {}, {}, {}, {}, None
    if case == "repeats":
        kw = dict(repeat_frac=0.6, repeat_len=(200, 800), repeat_copies=(50, 400), repeat_div=0.02)
    if case == "indels_flag_all":
        rkw = dict(sub_rate=0.02, indel_frac=0.9); pover = dict(flag_all=1, T=20)
    if case == "scoring":
        scoring = (2, 5, 7, 2); over = dict(a=2, b=5, o_del=7, e_del=2, o_ins=7, e_ins=2); pover = dict(T=40)
    g = synth.make_genome(500_000, seed=13, **kw)
    idx = fmindex.build_fmd_index(g)
    reads, _ = synth.make_reads(g, 2500, 150, seed=4, **rkw)
    if case == "indels_flag_all":
        # reads with a long deletion: two colinear regions whose merge goes through the patch test's global alignment
        rng = np.random.default_rng(5)
        for i in range(0, 600):
            p0 = int(rng.integers(0, len(g) - 400)); d = int(rng.integers(12, 60))
            x = np.concatenate([g[p0:p0 + 75], g[p0 + 75 + d:p0 + 150 + d]])
            reads[i] = x if i & 1 else synth.revcomp(x)
    flat, offs, lens = common.flat_reads(reads)
    s = oracle.seed_reads(oracle.fmd(idx), flat, offs, lens, 19, n_threads=4)
    co = ChainOpt(); L.bmh_chain_opt_default(C.byref(co))
    for k, v in over.items():
        setattr(co, k, v)
    ep = ExtParams.default()
    if scoring:
        ep = ExtParams(scoring[0], scoring[1], scoring[2], scoring[3], scoring[2], scoring[3], 0, 5)
    hj = HostJobs(g, flat, offs, lens, s, n_threads=4, opt=co)
    import oracle_py
    kp = oracle_py.KswParams(ep.a, ep.b, ep.o_del, ep.e_del, ep.o_ins, ep.e_ins, 0, 5, 1)
    o3, _, _ = oracle.extend_batch(*hj.jobs(), params=kp, n_threads=4)
    regs = np.ascontiguousarray(hj.merge(o3)); rpr = np.ascontiguousarray(hj.regs_per_read.copy()); fr = np.ascontiguousarray(hj.frac_rep(), dtype=np.float32)
    po = PostOpt(); L.bmh_post_opt_default(C.byref(po)); po.id0 = 77
    for k, v in pover.items():
        setattr(po, k, v)
    (h_out, h_opr), (c_out, c_opr) = both(lib, L, co, ep, po, g, pack(g), np.ascontiguousarray(flat), np.ascontiguousarray(offs), regs, rpr, fr)
    assert np.array_equal(h_opr, c_opr)
    assert np.array_equal(h_out, c_out), np.nonzero((h_out != c_out).any(1))[0][:5]
    assert len(h_out) > 2000 and (h_out[:, 12] >= 0).sum() > (50 if case != "plain" else 0)
    if case == "indels_flag_all":
        assert len(h_out) < len(regs) - 100          # regions were merged / dropped
    hj.free()


@pytest.mark.parametrize("flag_all", [0, 1])
def test_region_tail_core_with_alt_contigs_equals_host_form(oracle, flag_all):
    """ALT contigs (two-round primary marking, secondary_all, alt_sc, the rules of mem_reg2sam for hits on them): the device core with the table against
    bmh_finalize_regs with the table, on a repeat-rich genome of four sequences two of which are ALT copies of stretches of the others."""
    L = load_library()
    lib = core_lib()
    rng = np.random.default_rng(21)
    base = synth.make_genome(360_000, seed=17, repeat_frac=0.4, repeat_len=(150, 600), repeat_copies=(10, 80), repeat_div=0.02)
    alt1 = base[40_000:70_000].copy(); alt2 = base[200_000:220_000].copy()
    for a_ in (alt1, alt2):                                       # a diverged copy: a substitution every ~60 bases
        m = rng.random(len(a_)) < 1 / 60; a_[m] = (a_[m] + rng.integers(1, 4, int(m.sum()))) & 3
    g = np.ascontiguousarray(np.concatenate([base, alt1, alt2]))
    contigs = [("c1", 180_000), ("c2", 180_000), ("c1_alt", len(alt1)), ("c2_alt", len(alt2))]
    is_alt = np.ascontiguousarray([0, 0, 1, 1], dtype=np.uint8)
    idx = fmindex.build_fmd_index(g)
    reads, _ = synth.make_reads(g, 3000, 150, seed=6)
    reads[:1200] = synth.make_reads(g[40_000:70_000], 1200, 150, seed=7)[0]      # reads of a stretch that has an ALT copy
    flat, offs, lens = common.flat_reads(reads)
    s = oracle.seed_reads(oracle.fmd(idx), flat, offs, lens, 19, n_threads=4)
    co = ChainOpt(); L.bmh_chain_opt_default(C.byref(co)); co.contig_is_alt = is_alt.ctypes.data_as(C.c_void_p).value
    ep = ExtParams.default()
    hj = HostJobs(g, flat, offs, lens, s, n_threads=4, opt=co, contigs=contigs)
    o3, _, _ = oracle.extend_batch(*hj.jobs(), n_threads=4)
    regs = np.ascontiguousarray(hj.merge(o3)); rpr = np.ascontiguousarray(hj.regs_per_read.copy()); fr = np.ascontiguousarray(hj.frac_rep(), dtype=np.float32)
    po = PostOpt(); L.bmh_post_opt_default(C.byref(po)); po.id0 = 12; po.flag_all = flag_all; po.contig_is_alt = is_alt.ctypes.data_as(C.c_void_p).value
    (h_out, h_opr), (c_out, c_opr) = both(lib, L, co, ep, po, g, pack(g), np.ascontiguousarray(flat), np.ascontiguousarray(offs), regs, rpr, fr, contigs=contigs)
    assert np.array_equal(h_opr, c_opr)
    assert np.array_equal(h_out, c_out), (np.nonzero((h_out != c_out).any(1))[0][:5], h_out[(h_out != c_out).any(1)][:3], c_out[(h_out != c_out).any(1)][:3])
    assert (h_out[:, 15] & 2).sum() > 300 and (h_out[:, 15] >> 2 > 0).sum() > 100 and (h_out[:, 12] == 0x7FFFFFFF).sum() > 50      # ALT hits, shadowed hits, ALT hits with a parent
    hj.free()
