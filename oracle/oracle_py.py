"""ctypes bindings to the CPU checker -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

liboracle.so  : our C restatement (oracle/fmd_oracle.c, oracle/ksw_oracle.c)
_ref/libref.so: the reference's own compiled C + oracle/ref_harness.c (optional)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_u8p, _u32p, _u64p, _i32p = (C.POINTER(t) for t in (C.c_uint8, C.c_uint32, C.c_uint64, C.c_int32))


def build(ref: bool = True) -> None:
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    if ref and os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


class FmdT(C.Structure):
    _fields_ = [("primary", C.c_uint64), ("L2", C.c_uint64 * 5), ("seq_len", C.c_uint64),
                ("n_words", C.c_uint64), ("bwt", _u32p), ("sa_intv", C.c_int), ("n_sa", C.c_uint64),
                ("sa", _u32p), ("sa_bits", _u32p)]


class WorkT(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_blk", "n_sa", "n_fwd_steps", "n_back_steps", "n_lf_steps", "n_blk_fwd", "n_blk_back", "n_blk_lf")]


class SeedsT(C.Structure):
    _fields_ = [("n_seeds", C.c_uint64), ("rbeg", _u64p), ("qbeg", _i32p), ("score", _u32p),
                ("n_ref_pos", _u32p), ("prefix", _u32p), ("n_smems", C.c_uint64), ("smem_k", _u64p),
                ("smem_s", _u32p), ("smem_qb", _i32p), ("smem_qe", _i32p), ("smem_read", _u32p),
                ("work", WorkT)]


class KswParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("a", "b", "o_del", "e_del", "o_ins", "e_ins", "zdrop", "end_bonus", "n_penalty")]


def default_params(zdrop: int = 0) -> KswParams:
    # defaults of the GPU pipeline: src/bwamem.c:101-146 (a1 b4 o6 e1 zdrop0 clip5)
    return KswParams(1, 4, 6, 1, 6, 1, zdrop, 5, 1)


def _ptr(a, t):
    return a.ctypes.data_as(t)


class Oracle:
    def __init__(self):
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build(ref=False)
        self.lib = L = C.CDLL(path)
        L.oracle_seed_reads.restype = C.POINTER(SeedsT)
        L.oracle_seed_reads.argtypes = [C.POINTER(FmdT), _u8p, _u64p, _u32p, C.c_uint32, C.c_int, C.c_int]
        L.oracle_seeds_free.argtypes = [C.POINTER(SeedsT)]
        L.oracle_extend_batch.restype = C.c_uint64
        L.oracle_extend_batch.argtypes = [C.c_uint32, _u8p, _u32p, _u32p, _u8p, _u32p, _u32p, _u32p,
                                          C.POINTER(KswParams), _i32p, _i32p, C.c_int]
        L.fmd_occ.restype = C.c_uint64
        L.fmd_occ.argtypes = [C.POINTER(FmdT), C.c_uint64, C.c_int]
        L.fmd_inv_psi.restype = C.c_uint64
        L.fmd_inv_psi.argtypes = [C.POINTER(FmdT), C.c_uint64]
        L.fmd_sa.restype = C.c_uint64
        L.fmd_sa.argtypes = [C.POINTER(FmdT), C.c_uint64, C.c_void_p]
        L.oracle_ksw_global2.restype = C.c_int
        L.oracle_ksw_global2.argtypes = [C.c_int, _u8p, C.c_int, _u8p, C.POINTER(KswParams), C.c_int, _i32p, _u32p, C.c_int]
        L.oracle_reg2aln.restype = C.c_int
        L.oracle_reg2aln.argtypes = [C.POINTER(KswParams), C.c_int, C.c_int64, _u8p, C.c_int, _u8p, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                     C.c_int, C.c_int, C.POINTER(C.c_int64), _i32p, _i32p, _u32p, C.c_int, _i32p, C.c_char_p, C.c_int, _i32p]

    def global2(self, query, target, w, params: "KswParams | None" = None, cap: int = 1024):
        """oracle_ksw_global2 -> (score, cigar ops uint32[len<<4|op])"""
        p = params or default_params()
        q = np.ascontiguousarray(query, dtype=np.uint8); t = np.ascontiguousarray(target, dtype=np.uint8)
        n = np.zeros(1, np.int32); cg = np.zeros(cap, np.uint32)
        sc = self.lib.oracle_ksw_global2(len(q), _ptr(q, _u8p), len(t), _ptr(t, _u8p), C.byref(p), int(w), _ptr(n, _i32p), _ptr(cg, _u32p), cap)
        return int(sc), cg[: int(n[0])].copy()

    def reg2aln(self, pac, l_pac, read, qb, qe, rb, re, truesc, reg_w=300, opt_w=300, params: "KswParams | None" = None, cap: int = 512):
        """oracle_reg2aln -> dict(pos, is_rev, cigar, NM, MD, score)"""
        p = params or default_params()
        rd = np.ascontiguousarray(read, dtype=np.uint8)
        pos = C.c_int64(); isr = np.zeros(1, np.int32); n = np.zeros(1, np.int32); nm = np.zeros(1, np.int32); sc = np.zeros(1, np.int32)
        cg = np.zeros(cap, np.uint32); md = C.create_string_buffer(1024)
        self.lib.oracle_reg2aln(C.byref(p), opt_w, l_pac, _ptr(pac, _u8p), len(rd), _ptr(rd, _u8p), int(qb), int(qe), int(rb), int(re), int(truesc),
                                int(reg_w), C.byref(pos), _ptr(isr, _i32p), _ptr(n, _i32p), _ptr(cg, _u32p), cap, _ptr(nm, _i32p), md, 1024, _ptr(sc, _i32p))
        return dict(pos=int(pos.value), is_rev=int(isr[0]), cigar=cg[: int(n[0])].copy(), NM=int(nm[0]), MD=md.value.decode(), score=int(sc[0]))

    def fmd(self, idx) -> FmdT:
        """idx: bwamem_hip.fmindex.FMDIndex (arrays are kept alive on the returned struct)."""
        f = FmdT()
        f.primary = idx.primary
        for i in range(5):
            f.L2[i] = int(idx.L2[i])
        f.seq_len = idx.seq_len
        keep = [np.ascontiguousarray(idx.bwt_words, dtype=np.uint32), np.ascontiguousarray(idx.sa, dtype=np.uint32),
                np.ascontiguousarray(idx.sa_bits, dtype=np.uint32)]
        f.n_words = keep[0].shape[0]
        f.bwt = _ptr(keep[0], _u32p)
        f.sa_intv = idx.sa_intv
        f.n_sa = idx.n_sa
        f.sa = _ptr(keep[1], _u32p)
        f.sa_bits = _ptr(keep[2], _u32p)
        f._keep = keep
        return f

    def seed_reads(self, f: FmdT, reads: np.ndarray, offs: np.ndarray, lens: np.ndarray,
                   min_seed_len: int = 19, n_threads: int = 1) -> dict:
        reads = np.ascontiguousarray(reads, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        n = lens.shape[0]
        p = self.lib.oracle_seed_reads(C.byref(f), _ptr(reads, _u8p), _ptr(offs, _u64p), _ptr(lens, _u32p),
                                       n, min_seed_len, n_threads)
        s = p.contents
        ns, nm = int(s.n_seeds), int(s.n_smems)

        def arr(ptr, cnt, dt):
            return np.ctypeslib.as_array(ptr, shape=(max(cnt, 1),))[:cnt].astype(dt, copy=True)
        out = dict(
            rbeg=arr(s.rbeg, ns, np.uint64), qbeg=arr(s.qbeg, 2 * ns, np.int32).reshape(-1, 2),
            score=arr(s.score, ns, np.uint32), n_ref_pos=arr(s.n_ref_pos, n, np.uint32),
            prefix=arr(s.prefix, n, np.uint32), smem_k=arr(s.smem_k, nm, np.uint64),
            smem_s=arr(s.smem_s, nm, np.uint32), smem_qb=arr(s.smem_qb, nm, np.int32),
            smem_qe=arr(s.smem_qe, nm, np.int32), smem_read=arr(s.smem_read, nm, np.uint32),
            work={k: int(getattr(s.work, k)) for k, _ in WorkT._fields_})
        self.lib.oracle_seeds_free(p)
        return out

    def extend_batch(self, q, qoff, qlen, t, toff, tlen, h0, params: KswParams | None = None,
                     n_threads: int = 1, want_raw: bool = False):
        params = params or default_params()
        q = np.ascontiguousarray(q, dtype=np.uint8); t = np.ascontiguousarray(t, dtype=np.uint8)
        qoff, qlen, toff, tlen, h0 = (np.ascontiguousarray(x, dtype=np.uint32) for x in (qoff, qlen, toff, tlen, h0))
        n = qlen.shape[0]
        out3 = np.zeros((n, 3), dtype=np.int32)
        raw6 = np.zeros((n, 6), dtype=np.int32) if want_raw else None
        cells = self.lib.oracle_extend_batch(n, _ptr(q, _u8p), _ptr(qoff, _u32p), _ptr(qlen, _u32p), _ptr(t, _u8p),
                                             _ptr(toff, _u32p), _ptr(tlen, _u32p), _ptr(h0, _u32p), C.byref(params),
                                             _ptr(out3, _i32p), _ptr(raw6, _i32p) if want_raw else None, n_threads)
        return out3, raw6, int(cells)


class Ref:
    """The reference's own compiled C (oracle/_ref/libref.so). available() is False where it was not built."""

    PATH = os.path.join(_HERE, "_ref", "libref.so")

    @classmethod
    def available(cls) -> bool:
        return os.path.exists(cls.PATH)

    def __init__(self):
        self.lib = L = C.CDLL(self.PATH)
        L.ref_bwt_from_symbols.restype = C.c_void_p
        L.ref_bwt_from_symbols.argtypes = [_u8p, C.c_uint64, C.c_uint64, _u64p, C.c_int]
        L.ref_bwt_free.argtypes = [C.c_void_p]
        if hasattr(L, "ref_bwt_from_gpu_layout"):
            L.ref_bwt_from_gpu_layout.restype = C.c_void_p
            L.ref_bwt_from_gpu_layout.argtypes = [_u32p, C.c_uint64, C.c_uint64, _u64p, C.c_int, _u32p, _u32p]
        L.ref_occ.restype = C.c_uint64
        L.ref_occ.argtypes = [C.c_void_p, C.c_uint64, C.c_int]
        L.ref_sa.restype = C.c_uint64
        L.ref_sa.argtypes = [C.c_void_p, C.c_uint64]
        L.ref_collect_smems.restype = C.c_void_p
        L.ref_collect_smems.argtypes = [C.c_void_p, _u8p, _u64p, _u32p, C.c_uint32, C.c_int]
        L.ref_smems_n.restype = C.c_uint64
        L.ref_smems_n.argtypes = [C.c_void_p]
        L.ref_smems_get.argtypes = [C.c_void_p, _u64p, _u32p, _i32p, _i32p, _u32p]
        L.ref_smems_free.argtypes = [C.c_void_p]
        L.ref_locate.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, _u64p]
        L.ref_extend_batch.argtypes = [C.c_uint32, _u8p, _u32p, _u32p, _u8p, _u32p, _u32p, _u32p] + [C.c_int] * 9 + [_i32p, _i32p]
        if hasattr(L, "ref_global2"):
            L.ref_global2.restype = C.c_int
            L.ref_global2.argtypes = [C.c_int, _u8p, C.c_int, _u8p] + [C.c_int] * 7 + [_i32p, _u32p, C.c_int]

    def global2(self, query, target, w, params: "KswParams | None" = None, cap: int = 1024):
        """the reference's ksw_global2 -> (score, cigar ops)"""
        p = params or default_params()
        q = np.ascontiguousarray(query, dtype=np.uint8); t = np.ascontiguousarray(target, dtype=np.uint8)
        sc = np.zeros(1, np.int32); cg = np.zeros(cap, np.uint32)
        assert p.n_penalty == 1
        n = self.lib.ref_global2(len(q), _ptr(q, _u8p), len(t), _ptr(t, _u8p), p.a, p.b, p.o_del, p.e_del, p.o_ins, p.e_ins, int(w),
                                 _ptr(sc, _i32p), _ptr(cg, _u32p), cap)
        return int(sc[0]), cg[:n].copy()

    def bwt_from_index(self, idx):
        """Vanilla-layout bwt_t built from the BWT symbols of a GPU-layout FMDIndex."""
        sym = bwt_symbols(idx)
        L2 = np.ascontiguousarray(idx.L2, dtype=np.uint64)
        return self.lib.ref_bwt_from_symbols(_ptr(sym, _u8p), idx.seq_len, idx.primary, _ptr(L2, _u64p), idx.sa_intv)

    def bwt_from_index_fast(self, idx):
        """Vanilla-layout bwt_t by re-interleaving the GPU-layout words and copying the suffix-array samples (hg38-scale texts)."""
        w = np.ascontiguousarray(idx.bwt_words, dtype=np.uint32)
        nblk = (idx.seq_len + 63) // 64
        if w.shape[0] < (nblk + 1) * 8:                     # file layout: a short last block; pad to whole blocks
            w = np.concatenate([w[: w.shape[0] - 4], np.zeros((nblk + 1) * 8 - (w.shape[0] - 4), np.uint32)])
        L2 = np.ascontiguousarray(idx.L2, dtype=np.uint64)
        sa = np.ascontiguousarray(idx.sa, dtype=np.uint32); bits = np.ascontiguousarray(idx.sa_bits, dtype=np.uint32)
        return self.lib.ref_bwt_from_gpu_layout(_ptr(w, _u32p), idx.seq_len, idx.primary, _ptr(L2, _u64p), idx.sa_intv, _ptr(sa, _u32p), _ptr(bits, _u32p))

    def seed_reads(self, bwt, reads, offs, lens, min_seed_len=19) -> dict:
        reads = np.ascontiguousarray(reads, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        n = lens.shape[0]
        h = self.lib.ref_collect_smems(bwt, _ptr(reads, _u8p), _ptr(offs, _u64p), _ptr(lens, _u32p), n, min_seed_len)
        nm = int(self.lib.ref_smems_n(h))
        k = np.zeros(nm, np.uint64); s = np.zeros(nm, np.uint32)
        qb = np.zeros(nm, np.int32); qe = np.zeros(nm, np.int32); rd = np.zeros(nm, np.uint32)
        if nm:
            self.lib.ref_smems_get(h, _ptr(k, _u64p), _ptr(s, _u32p), _ptr(qb, _i32p), _ptr(qe, _i32p), _ptr(rd, _u32p))
        self.lib.ref_smems_free(h)
        ns = int(s.sum())
        rbeg = np.zeros(ns, np.uint64); qbeg = np.zeros((ns, 2), np.int32); score = np.zeros(ns, np.uint32)
        n_ref = np.zeros(n, np.uint32)
        o = 0
        for i in range(nm):
            si = int(s[i])
            self.lib.ref_locate(bwt, int(k[i]), si, _ptr(rbeg[o:], _u64p))
            qbeg[o:o + si, 0] = qb[i]; qbeg[o:o + si, 1] = qe[i]
            score[o] = si
            n_ref[rd[i]] += si
            o += si
        prefix = np.zeros(n, np.uint32)
        if n:
            prefix[1:] = np.cumsum(n_ref)[:-1]
        return dict(rbeg=rbeg, qbeg=qbeg, score=score, n_ref_pos=n_ref, prefix=prefix, smem_k=k, smem_s=s,
                    smem_qb=qb, smem_qe=qe, smem_read=rd)

    def extend_batch(self, q, qoff, qlen, t, toff, tlen, h0, params: KswParams | None = None, w: int = 300):
        p = params or default_params()
        q = np.ascontiguousarray(q, dtype=np.uint8); t = np.ascontiguousarray(t, dtype=np.uint8)
        qoff, qlen, toff, tlen, h0 = (np.ascontiguousarray(x, dtype=np.uint32) for x in (qoff, qlen, toff, tlen, h0))
        n = qlen.shape[0]
        out3 = np.zeros((n, 3), np.int32); raw6 = np.zeros((n, 6), np.int32)
        assert p.n_penalty == 1
        self.lib.ref_extend_batch(n, _ptr(q, _u8p), _ptr(qoff, _u32p), _ptr(qlen, _u32p), _ptr(t, _u8p), _ptr(toff, _u32p),
                                  _ptr(tlen, _u32p), _ptr(h0, _u32p), p.a, p.b, p.o_del, p.e_del, p.o_ins, p.e_ins, w,
                                  p.zdrop, p.end_bonus, _ptr(out3, _i32p), _ptr(raw6, _i32p))
        return out3, raw6


def bwt_symbols(idx) -> np.ndarray:
    """Unpack the BWT symbol string (codes 0..3, length seq_len) from the GPU-layout words."""
    n = idx.seq_len
    nblk = (n + 63) // 64
    w = idx.bwt_words
    full = np.zeros(nblk * 8, dtype=np.uint32)
    full[:min(w.shape[0], nblk * 8)] = w[:nblk * 8]
    words = full.reshape(nblk, 8)[:, 4:].reshape(-1)
    sh = (30 - 2 * np.arange(16)).astype(np.uint32)
    sym = ((words[:, None] >> sh[None, :]) & 3).astype(np.uint8).reshape(-1)
    return np.ascontiguousarray(sym[:n])
