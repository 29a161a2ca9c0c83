"""ctypes binding of libbwamem_hip.so (include/bwamem_hip.h, include/seed_gen.h)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_u8p, _u32p, _u64p, _i32p = (C.POINTER(t) for t in (C.c_uint8, C.c_uint32, C.c_uint64, C.c_int32))


class HipLibraryMissing(RuntimeError):
    pass


def lib_path() -> str:
    # BMH_LIB: another build of the same library (A/B runs of kernel variants, scripts/build_variants.sh)
    return os.environ.get("BMH_LIB") or os.path.join(_PKG, "libbwamem_hip.so")


# every symbol include/bwamem_hip.h and include/seed_gen.h declare
EXPORTED_SYMBOLS = [
    "bmh_last_error", "bmh_device_count", "bmh_set_device", "bmh_index_upload", "bmh_index_from_device",
    "bmh_index_free", "bmh_index_probe", "bmh_index_replicate", "bmh_rccl_where", "bmh_rccl_unique_id", "bmh_rccl_comm_init_rank", "bmh_rccl_comm_destroy", "bmh_index_broadcast_rccl", "bmh_index_replicate_all", "bmh_shard_range", "bmh_index_densify_sa", "bmh_index_build", "bmh_seed_ws_create", "bmh_seed_ws_free", "bmh_seed_batch", "bmh_seed_last_timing",
    "bmh_host_pin", "bmh_host_unpin", "bmh_extend_batch", "bmh_extend_last_ms", "bmh_extend_last_unsupported", "bmh_extend_set_packed", "bmh_tune_set", "bmh_wtrace_start", "bmh_wtrace_stop", "bmh_wtrace_kept", "bmh_extend_release", "bmh_finalize_release", "bmh_matesw_release", "bmh_calib_gather", "bmh_calib_valu", "bmh_calib_valu_placed", "bmh_calib_last_clock",
    "bmh_jobs_frac_rep", "bmh_post_opt_default", "bmh_finalize_regs", "bmh_finalize_regs_device", "bmh_finalize_regs_device_last_ms", "bmh_sam_need_cigar", "bmh_format_sam", "bmh_free",
    "bmh_pe_opt_default", "bmh_finalize_pairs", "bmh_finalize_pairs_dev", "bmh_dedup_regs_device", "bmh_finalize_pairs_deduped", "bmh_rescue_check_counts", "bmh_sam_need_cigar_pe", "bmh_format_sam_pe",
    "bmh_chain_opt_default", "bmh_chain_last_timing", "bmh_build_jobs", "bmh_jobs_free", "bmh_jobs_sizes", "bmh_jobs_arrays", "bmh_merge_regs",
    "bmh_chain_ws_create", "bmh_chain_ws_free", "bmh_chain_set_contigs", "bmh_chain_set_alt", "bmh_effective_cpus", "bmh_aligner_create", "bmh_aligner_free", "bmh_aligner_run", "bmh_aligner_run_fasta", "bmh_chain_set_materialize", "bmh_chain_batch",
    "bmh_chain_extend", "bmh_chain_merge", "bmh_chain_extend_merge", "bmh_chain_extend_merge_timing", "bmh_cigar_batch", "bmh_cigar_release",
    "bmh_sam_select_work", "bmh_sam_select_device", "bmh_cigar_pack_work", "bmh_cigar_pack_sizes", "bmh_cigar_pack", "bmh_sam_text_work", "bmh_sam_text_sizes", "bmh_sam_text_write", "bmh_sam_text_check",
    "bwt_destroy_gpu", "bwt_restore_sa_gpu", "bwt_restore_bwt_gpu", "gpu_cpy_wrapper",
    "pre_calc_seed_intervals_wrapper", "free_gpuseed_data", "seed_gpu", "seed_gpu_last_n_reads",
    "bmh_reads_load_fasta", "bmh_reads_free", "bmh_fasta_scan",
]


class SamDev(C.Structure):
    """bmh_sam_dev_t: device pointers of bmh_sam_text_sizes / bmh_sam_text_write"""
    _fields_ = [("n_reads", C.c_uint32), ("d_names", C.c_void_p), ("d_name_off", C.c_void_p), ("d_reads", C.c_void_p), ("d_offs", C.c_void_p), ("d_lens", C.c_void_p),
                ("n_contigs", C.c_int), ("d_contig_names", C.c_void_p), ("d_contig_name_off", C.c_void_p), ("d_contig_offset", C.c_void_p),
                ("d_fin", C.c_void_p), ("d_fin_per_read", C.c_void_p), ("d_slot", C.c_void_p), ("d_aln", C.c_void_p), ("d_cig_off", C.c_void_p), ("d_packed", C.c_void_p),
                ("d_h_rec", C.c_void_p), ("d_unflag", C.c_void_p)]


class ReadSetT(C.Structure):
    """bmh_read_set_t"""
    _fields_ = [("n_reads", C.c_uint64), ("n_bases", C.c_uint64), ("n_name_bytes", C.c_uint64), ("ascii", C.c_void_p), ("codes", C.c_void_p),
                ("offs", C.c_void_p), ("lens", C.c_void_p), ("names", C.c_void_p), ("name_offs", C.c_void_p)]


class _ReadSetOwner:
    """keeps a bmh_read_set_t alive for the numpy arrays that look into it; frees it with the last of them"""

    def __init__(self, L, rs):
        self.L, self.rs = L, rs

    def __del__(self):
        try:
            self.L.bmh_reads_free(C.byref(self.rs))
        except Exception:
            pass


def fasta_scan(path: str, n_threads: int = 0) -> dict:
    """bmh_fasta_scan: reads / bases / name bytes / longest read of a read file, from one counting pass (nothing is loaded)"""
    L = load_library()
    L.bmh_fasta_scan.restype = C.c_int
    L.bmh_fasta_scan.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_uint64)]
    out = (C.c_uint64 * 4)()
    rc = L.bmh_fasta_scan(path.encode(), n_threads, out)
    if rc != 0:
        raise RuntimeError(f"bmh_fasta_scan rc={rc}: " + _err(L))
    return dict(n_reads=int(out[0]), n_bases=int(out[1]), n_name_bytes=int(out[2]), max_len=int(out[3]))


def load_fasta_reads(path: str, n_threads: int = 0) -> dict:
    """bmh_reads_load_fasta: the read file as numpy arrays over the library's own arrays (no copies; freed with the last array)"""
    L = load_library()
    rs = ReadSetT()
    L.bmh_reads_load_fasta.argtypes = [C.c_char_p, C.c_int, C.POINTER(ReadSetT)]
    L.bmh_reads_free.argtypes = [C.POINTER(ReadSetT)]
    if L.bmh_reads_load_fasta(path.encode(), n_threads, C.byref(rs)) != 0:
        msg = _err(L)
        raise ValueError(msg) if "expected alternating" in msg else RuntimeError("bmh_reads_load_fasta: " + msg)
    owner = _ReadSetOwner(L, rs)

    def arr(ptr, n, dt):
        n = int(n)
        if n == 0 or not ptr:
            return np.zeros(0, dt)
        buf = (C.c_uint8 * (n * np.dtype(dt).itemsize)).from_address(ptr)
        buf._owner = owner
        return np.frombuffer(buf, dtype=dt)
    return dict(ascii=arr(rs.ascii, rs.n_bases, np.uint8), codes=arr(rs.codes, rs.n_bases, np.uint8), offs=arr(rs.offs, rs.n_reads, np.uint64),
                lens=arr(rs.lens, rs.n_reads, np.uint32), names=arr(rs.names, rs.n_name_bytes, np.uint8), name_offs=arr(rs.name_offs, rs.n_reads, np.uint64))


class AlignStats(C.Structure):
    """bmh_align_stats_t"""
    _fields_ = [("n_reads", C.c_uint64), ("n_bytes", C.c_uint64), ("n_batches", C.c_uint32), ("n_lanes", C.c_int)] + \
               [(n, C.c_double) for n in ("seconds", "format_seconds", "h2d_seconds", "seed_seconds", "chain_extend_seconds", "tail_seconds", "select_seconds", "cigar_seconds", "gate_wait_seconds", "h2d_copy_seconds", "d2h_copy_seconds")] + \
               [("h2d_bytes", C.c_uint64), ("d2h_bytes", C.c_uint64)]


SAM_SINK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)


class NativeAligner:
    """bmh_aligner_t: the native reads -> SAM pipeline; lanes (streams, workspaces, pinned staging) are kept between runs"""

    def __init__(self, index: "Index", pac: np.ndarray, l_pac: int, contigs, is_alt, copt, ep, po, pe):
        L = load_library()
        L.bmh_aligner_create.restype = C.c_void_p
        L.bmh_aligner_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_char_p), C.c_void_p, C.c_void_p, C.POINTER(ChainOpt), C.POINTER(ExtParams),
                                         C.POINTER(PostOpt), C.POINTER(PeOpt)]
        L.bmh_aligner_free.argtypes = [C.c_void_p]
        L.bmh_aligner_run.restype = C.c_int
        L.bmh_aligner_run.argtypes = [C.c_void_p, C.POINTER(ReadSetT), C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int, SAM_SINK, C.c_void_p, C.POINTER(AlignStats)]
        names = (C.c_char_p * len(contigs))(*[c[0].encode() for c in contigs])
        lens = np.ascontiguousarray([c[1] for c in contigs], dtype=np.int32)
        alt = np.ascontiguousarray(is_alt, dtype=np.uint8) if is_alt is not None else None
        self._pac = np.ascontiguousarray(pac, dtype=np.uint8)          # (borrowed by the handle: kept alive here)
        self._index = index
        self.options = (bytes(copt), bytes(ep), bytes(po), bytes(pe))  # what the handle was made with: the caller makes a new one when they change
        self.handle = L.bmh_aligner_create(index.handle, self._pac.ctypes.data, int(l_pac), len(contigs), names, lens.ctypes.data, alt.ctypes.data if alt is not None else None,
                                           C.byref(copt), C.byref(ep), C.byref(po), C.byref(pe))
        if not self.handle:
            raise RuntimeError("bmh_aligner_create: " + _err(L))

    def run(self, rs, cuts, paired: bool, write, n_lanes: int = 2, n_threads: int = 0) -> "AlignStats":
        """the batches [cuts[b], cuts[b+1]) of the read set `rs` (an aligner.ReadSet; codes required); write(memoryview) receives every
        batch's SAM records in order"""
        L = load_library()
        keep = [np.ascontiguousarray(rs.ascii, dtype=np.uint8), np.ascontiguousarray(rs.codes, dtype=np.uint8), np.ascontiguousarray(rs.offs, dtype=np.uint64),
                np.ascontiguousarray(rs.lens, dtype=np.uint32), np.ascontiguousarray(rs.name_blob, dtype=np.uint8), np.ascontiguousarray(rs.name_off, dtype=np.uint64)]
        c = ReadSetT()
        c.n_reads = len(keep[3]); c.n_bases = int(keep[2][-1] + keep[3][-1]) if len(keep[3]) else 0; c.n_name_bytes = len(keep[4])
        c.ascii, c.codes, c.offs, c.lens, c.names, c.name_offs = (k.ctypes.data for k in keep)
        cu = np.ascontiguousarray(cuts, dtype=np.uint64)
        err = []

        def sink(_user, ptr, n):
            try:
                write(memoryview((C.c_char * n).from_address(ptr)))
                return 0
            except BaseException as e:                      # noqa: BLE001 -- reported after the run (an exception must not cross the C frames)
                err.append(e)
                return 1
        cb = SAM_SINK(sink)
        st = AlignStats()
        rc = L.bmh_aligner_run(self.handle, C.byref(c), cu.ctypes.data, len(cu) - 1, 1 if paired else 0, int(n_lanes), int(n_threads), cb, None, C.byref(st))
        if err:
            raise err[0]
        if rc != 0:
            raise (CapacityError if rc == -3 else RuntimeError)(f"bmh_aligner_run rc={rc}: " + _err(L))
        return st

    def run_fasta(self, path: str, paired: bool, write, batch_bases: int = 0, batch_reads: int = 0, n_lanes: int = 2, n_threads: int = 0) -> "AlignStats":
        """bmh_aligner_run_fasta: the read file `path` batch by batch (cut by bases like the reference's bseq_read, or by reads), a loader thread ahead of the
        lanes; write(memoryview) receives every batch's SAM records in order"""
        L = load_library()
        L.bmh_aligner_run_fasta.restype = C.c_int
        L.bmh_aligner_run_fasta.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, SAM_SINK, C.c_void_p, C.POINTER(AlignStats)]
        err = []

        def sink(_user, ptr, n):
            try:
                write(memoryview((C.c_char * n).from_address(ptr)))
                return 0
            except BaseException as e:                      # noqa: BLE001 -- reported after the run (an exception must not cross the C frames)
                err.append(e)
                return 1
        cb = SAM_SINK(sink)
        st = AlignStats()
        rc = L.bmh_aligner_run_fasta(self.handle, path.encode(), int(batch_bases), int(batch_reads), 1 if paired else 0, int(n_lanes), int(n_threads), cb, None, C.byref(st))
        if err:
            raise err[0]
        if rc != 0:
            raise (CapacityError if rc == -3 else RuntimeError)(f"bmh_aligner_run_fasta rc={rc}: " + _err(L))
        return st

    def free(self):
        if self.handle:
            load_library().bmh_aligner_free(self.handle)
            self.handle = None


class SeedsT(C.Structure):
    _fields_ = [("n_seeds", C.c_uint64), ("n_smems", C.c_uint64), ("n_cands", C.c_uint64),
                ("d_rbeg", C.c_void_p), ("d_qbeg", C.c_void_p), ("d_score", C.c_void_p),
                ("d_n_ref_pos", C.c_void_p), ("d_prefix", C.c_void_p)]


class ExtParams(C.Structure):
    """bmh_ext_params_t; defaults = the GPU pipeline's (src/bwamem.c:101-146)."""
    _fields_ = [(n, C.c_int) for n in ("a", "b", "o_del", "e_del", "o_ins", "e_ins", "zdrop", "end_bonus")]

    @classmethod
    def default(cls, zdrop: int = 0) -> "ExtParams":
        return cls(1, 4, 6, 1, 6, 1, zdrop, 5)


class ChainOpt(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("a", "b", "o_del", "e_del", "o_ins", "e_ins", "w", "min_seed_len", "max_occ",
                                       "max_chain_gap", "min_chain_weight", "max_chain_extend")] + \
               [("mask_level", C.c_float), ("drop_ratio", C.c_float), ("contig_is_alt", C.c_void_p)]


class PostOpt(C.Structure):
    """bmh_post_opt_t"""
    _fields_ = [("T", C.c_int), ("mask_level_redun", C.c_float), ("mapQ_coef_len", C.c_float), ("mapQ_coef_fac", C.c_int),
                ("flag_all", C.c_int), ("id0", C.c_int64), ("XA_drop_ratio", C.c_float), ("max_XA_hits", C.c_int),
                ("no_multi", C.c_int), ("softclip", C.c_int), ("max_XA_hits_alt", C.c_int), ("contig_is_alt", C.c_void_p), ("rg_id", C.c_char_p)]


class PeOpt(C.Structure):
    """bmh_pe_opt_t"""
    _fields_ = [("pen_unpaired", C.c_int), ("max_ins", C.c_int), ("max_matesw", C.c_int), ("no_rescue", C.c_int), ("no_pairing", C.c_int)]


class BuildStats(C.Structure):
    """bmh_build_stats_t"""
    _fields_ = [("round0_passes", C.c_int), ("doubling_rounds", C.c_int), ("verified", C.c_int), ("unresolved_after_round0", C.c_uint64),
                ("round0_seconds", C.c_double), ("sa_seconds", C.c_double), ("verify_seconds", C.c_double), ("total_seconds", C.c_double)]


class DevJobsT(C.Structure):
    """bmh_dev_jobs_t"""
    _fields_ = [(n, C.c_uint64) for n in ("n_jobs", "n_regs", "q_bytes", "t_bytes", "n_heavy_reads")] + \
               [(n, C.c_void_p) for n in ("d_q", "d_qoff", "d_qlen", "d_t", "d_toff", "d_tlen", "d_h0", "d_job_read", "d_job_reg",
                                          "d_job_side", "d_regs_per_read", "d_frac_rep")]


# mirrors of include/seed_gen.h
class BwtTGpu(C.Structure):
    _fields_ = [("primary", C.c_uint64), ("L2", _u64p), ("seq_len", C.c_uint64), ("bwt_size", C.c_uint64),
                ("bwt", _u32p), ("sa_intv", C.c_int), ("n_sa", C.c_uint64), ("sa", _u32p),
                ("sa_upper_bits", _u32p), ("pack_size", C.c_uint8)]


class MemSeedVGpu(C.Structure):
    _fields_ = [("rbeg", _u64p), ("qbeg", _i32p), ("score", _u32p), ("n_ref_pos_fow_rev_results", _u32p),
                ("n_ref_pos_fow_rev_prefix_sums", _u32p), ("file_bytes_skip", C.c_uint64)]


class GpuseedStorageVector(C.Structure):
    _fields_ = [("read_file", C.c_char_p), ("query_file", C.c_char_p), ("bwt", C.POINTER(BwtTGpu)),
                ("bwt_gpu", BwtTGpu), ("pre_calc_seed_intervals", C.c_void_p),
                ("pre_calc_seed_intervals_flag", C.c_int), ("pre_calc_seed_len", C.c_int),
                ("min_seed_size", C.c_int), ("is_smem", C.c_int), ("file_bytes_skip", C.c_uint64)]


_LIB = None


def sources_sha16() -> str:
    """First 16 hex digits of the sha256 over the library's sources (csrc/*.hip, *.h, *.cpp and include/**/*.h, in name order):
    the build id a counter profile under profiles/ is stamped with (scripts/summarize_profiles.py) and bench.py compares
    before it quotes a counter from such a file."""
    import glob
    import hashlib
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    root = os.path.dirname(here)
    files = sorted(glob.glob(os.path.join(here, "csrc", "*.hip")) + glob.glob(os.path.join(here, "csrc", "*.h")) + glob.glob(os.path.join(here, "csrc", "*.cpp")) +
                   glob.glob(os.path.join(root, "include", "**", "*.h"), recursive=True))
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.relpath(f, root).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def load_library() -> C.CDLL:
    """Load libbwamem_hip.so; raises HipLibraryMissing (never falls back) if it is not built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    p = lib_path()
    if not os.path.exists(p):
        raise HipLibraryMissing(f"{p} not found: build it with `make -C bwa-mem_gpu_amd/csrc` "
                                "(or __graft_entry__.build()); there is no CPU fallback")
    L = C.CDLL(p)
    L.bmh_last_error.restype = C.c_char_p
    L.bmh_device_count.restype = C.c_int
    L.bmh_set_device.argtypes = [C.c_int]
    L.bmh_index_upload.restype = C.c_void_p
    L.bmh_index_upload.argtypes = [C.c_uint64, _u64p, C.c_uint64, _u32p, C.c_uint64, C.c_int, _u32p, C.c_uint64,
                                   _u32p, _u8p, C.c_uint64]
    L.bmh_index_from_device.restype = C.c_void_p
    L.bmh_index_from_device.argtypes = [C.c_uint64, _u64p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p,
                                        C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64]
    L.bmh_index_free.argtypes = [C.c_void_p]
    L.bmh_tune_set.restype = C.c_int
    L.bmh_wtrace_start.restype = C.c_int
    L.bmh_wtrace_start.argtypes = [C.c_uint32]
    L.bmh_wtrace_stop.restype = C.c_int64
    L.bmh_wtrace_stop.argtypes = [C.c_void_p, C.c_uint32]
    L.bmh_wtrace_kept.restype = C.c_uint32
    L.bmh_tune_set.argtypes = [C.c_char_p, C.c_int, C.c_int]
    L.bmh_rccl_where.restype = C.c_char_p
    L.bmh_rccl_unique_id.restype = C.c_int
    L.bmh_rccl_unique_id.argtypes = [C.c_void_p]
    L.bmh_rccl_comm_init_rank.restype = C.c_int
    L.bmh_rccl_comm_init_rank.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_int]
    L.bmh_rccl_comm_destroy.argtypes = [C.c_void_p]
    L.bmh_index_broadcast_rccl.restype = C.c_int
    L.bmh_index_broadcast_rccl.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p]
    L.bmh_index_replicate_all.restype = C.c_int
    L.bmh_index_replicate_all.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
    L.bmh_index_probe.restype = C.c_int
    L.bmh_index_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p]
    L.bmh_index_densify_sa.restype = C.c_int
    L.bmh_index_densify_sa.argtypes = [C.c_void_p, C.c_int]
    L.bmh_index_build.restype = C.c_int
    L.bmh_index_build.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, _u64p, _u64p, C.c_int, C.POINTER(BuildStats)]
    L.bmh_seed_ws_create.restype = C.c_void_p
    L.bmh_seed_ws_create.argtypes = [C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64]
    L.bmh_seed_ws_free.argtypes = [C.c_void_p]
    L.bmh_seed_batch.restype = C.c_int
    L.bmh_seed_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int,
                                 C.c_void_p, C.POINTER(SeedsT)]
    L.bmh_seed_last_timing.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    L.bmh_extend_batch.restype = C.c_int
    L.bmh_extend_batch.argtypes = [C.c_void_p] * 7 + [C.c_uint32, C.POINTER(ExtParams), C.c_void_p, C.c_void_p, C.c_void_p]
    L.bmh_extend_last_ms.restype = C.c_float
    L.bmh_extend_last_unsupported.restype = C.c_int64
    L.bmh_calib_gather.restype = C.c_int
    L.bmh_calib_gather.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_float)]
    L.bmh_calib_valu.restype = C.c_int
    L.bmh_calib_valu.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_double)]
    L.bmh_calib_valu_placed.restype = C.c_int
    L.bmh_calib_valu_placed.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_double), C.c_void_p, C.POINTER(C.c_uint)]
    L.bmh_chain_opt_default.argtypes = [C.POINTER(ChainOpt)]
    L.bmh_build_jobs.restype = C.c_void_p
    L.bmh_build_jobs.argtypes = [C.POINTER(ChainOpt), C.c_int64, _u8p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, _u8p, _u64p, _u32p,
                                 _u64p, _i32p, _u32p, _u32p, _u32p, C.c_int]
    L.bmh_jobs_free.argtypes = [C.c_void_p]
    L.bmh_jobs_sizes.argtypes = [C.c_void_p, _u64p, _u64p, _u64p, _u64p]
    L.bmh_jobs_arrays.argtypes = [C.c_void_p] + [C.POINTER(C.c_void_p)] * 11
    L.bmh_jobs_frac_rep.restype = C.POINTER(C.c_float)
    L.bmh_jobs_frac_rep.argtypes = [C.c_void_p]
    L.bmh_post_opt_default.argtypes = [C.POINTER(PostOpt)]
    L.bmh_finalize_regs.restype = C.c_int64
    L.bmh_finalize_regs.argtypes = [C.POINTER(ChainOpt), C.POINTER(ExtParams), C.POINTER(PostOpt), C.c_int64, _u8p, C.c_uint32, _u8p, _u64p,
                                    _i32p, _u32p, C.POINTER(C.c_float), C.c_int, C.c_void_p, _i32p, _u32p, C.c_int]
    L.bmh_finalize_regs_device.restype = C.c_int64
    L.bmh_finalize_regs_device.argtypes = [C.c_void_p, C.POINTER(ChainOpt), C.POINTER(ExtParams), C.POINTER(PostOpt), C.c_void_p, C.c_void_p, C.c_uint32,
                                           C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.bmh_finalize_regs_device_last_ms.restype = C.c_float
    L.bmh_sam_need_cigar.restype = C.c_int64
    L.bmh_sam_need_cigar.argtypes = [C.POINTER(PostOpt), _i32p, _u32p, C.c_uint32, _u8p]
    L.bmh_format_sam.restype = C.c_void_p
    L.bmh_format_sam.argtypes = [C.POINTER(PostOpt), C.c_uint32, C.c_void_p, _u64p, _u8p, _u64p, _u32p, C.c_int, C.POINTER(C.c_char_p), C.c_void_p,
                                 _i32p, _u32p, C.c_void_p, _i32p, _u32p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_size_t)]
    L.bmh_free.argtypes = [C.c_void_p]
    L.bmh_pe_opt_default.argtypes = [C.POINTER(PeOpt)]
    L.bmh_finalize_pairs.restype = C.c_int64
    L.bmh_finalize_pairs.argtypes = [C.POINTER(ChainOpt), C.POINTER(ExtParams), C.POINTER(PostOpt), C.POINTER(PeOpt), C.c_int64, _u8p, C.c_uint32, _u8p, _u64p,
                                     _u32p, _i32p, _u32p, C.POINTER(C.c_float), C.c_int, C.c_void_p, C.c_void_p, _i32p, C.c_uint64, _u32p, _i32p, _i32p,
                                     C.c_void_p, C.c_int]
    L.bmh_finalize_pairs_dev.restype = C.c_int64
    L.bmh_finalize_pairs_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + list(L.bmh_finalize_pairs.argtypes)
    L.bmh_rescue_check_counts.restype = None
    L.bmh_rescue_check_counts.argtypes = [C.POINTER(C.c_uint64)]
    L.bmh_finalize_pairs_deduped.restype = C.c_int64
    L.bmh_finalize_pairs_deduped.argtypes = list(L.bmh_finalize_pairs_dev.argtypes)
    L.bmh_dedup_regs_device.restype = C.c_int64
    L.bmh_dedup_regs_device.argtypes = [C.c_void_p, C.POINTER(ChainOpt), C.POINTER(ExtParams), C.POINTER(PostOpt), C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p,
                                        C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.bmh_sam_need_cigar_pe.restype = C.c_int64
    L.bmh_sam_need_cigar_pe.argtypes = [C.POINTER(PostOpt), _i32p, _u32p, _i32p, C.c_uint32, _u8p]
    L.bmh_format_sam_pe.restype = C.c_void_p
    L.bmh_format_sam_pe.argtypes = [C.POINTER(PostOpt), C.c_uint32, C.c_void_p, _u64p, _u8p, _u64p, _u32p, C.c_int, C.POINTER(C.c_char_p), C.c_void_p,
                                    _i32p, _u32p, _i32p, _i32p, C.c_void_p, _i32p, _u32p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_size_t)]
    L.bmh_merge_regs.restype = C.c_int
    L.bmh_merge_regs.argtypes = [C.c_void_p, _i32p, _i32p]
    L.bmh_chain_ws_create.restype = C.c_void_p
    L.bmh_chain_ws_create.argtypes = [C.c_uint32, C.c_uint64]
    L.bmh_chain_ws_free.argtypes = [C.c_void_p]
    L.bmh_chain_set_contigs.restype = C.c_int
    L.bmh_chain_set_contigs.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.bmh_chain_set_alt.restype = C.c_int
    L.bmh_chain_set_alt.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.bmh_chain_batch.restype = C.c_int
    L.bmh_chain_batch.argtypes = [C.c_void_p, C.POINTER(ChainOpt), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                  C.POINTER(SeedsT), C.c_void_p, C.POINTER(DevJobsT)]
    L.bmh_chain_set_materialize.restype = C.c_int
    L.bmh_chain_set_materialize.argtypes = [C.c_void_p, C.c_int]
    L.bmh_chain_extend.restype = C.c_int
    L.bmh_chain_extend.argtypes = [C.c_void_p, C.POINTER(ExtParams), C.c_void_p, C.c_void_p, C.c_void_p]
    L.bmh_chain_extend_merge.restype = C.c_int
    L.bmh_chain_extend_merge.argtypes = [C.c_void_p, C.POINTER(ChainOpt), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                         C.POINTER(SeedsT), C.POINTER(ExtParams), C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(DevJobsT)]
    L.bmh_chain_extend_merge_timing.restype = C.c_int
    L.bmh_chain_extend_merge_timing.argtypes = [C.c_void_p, C.POINTER(C.c_float), _u64p]
    L.bmh_chain_last_timing.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    L.bmh_chain_merge.restype = C.c_int
    L.bmh_chain_merge.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.bmh_cigar_batch.restype = C.c_int
    L.bmh_cigar_batch.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_uint32, C.POINTER(ExtParams), C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_int, C.c_void_p, C.c_void_p]
    L.bmh_cigar_release.restype = None
    L.bmh_cigar_release.argtypes = [C.c_void_p]
    L.bmh_sam_select_work.restype = C.c_size_t
    L.bmh_sam_select_work.argtypes = [C.c_uint32, C.c_uint64]
    L.bmh_sam_select_device.restype = C.c_int64
    L.bmh_sam_select_device.argtypes = [C.POINTER(PostOpt), C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.bmh_cigar_pack_work.restype = C.c_size_t
    L.bmh_cigar_pack_work.argtypes = [C.c_uint32]
    L.bmh_cigar_pack_sizes.restype = C.c_int64
    L.bmh_cigar_pack_sizes.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.bmh_cigar_pack.restype = C.c_int
    L.bmh_cigar_pack.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.bmh_sam_text_work.restype = C.c_size_t
    L.bmh_sam_text_work.argtypes = [C.c_uint32]
    L.bmh_sam_text_sizes.restype = C.c_int64
    L.bmh_sam_text_sizes.argtypes = [C.POINTER(PostOpt), C.POINTER(SamDev), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.bmh_sam_text_write.restype = C.c_int
    L.bmh_sam_text_write.argtypes = [C.POINTER(PostOpt), C.POINTER(SamDev), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.bmh_sam_text_check.restype = C.c_int
    L.bmh_sam_text_check.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    L.bwt_restore_bwt_gpu.restype = C.POINTER(BwtTGpu)
    L.bwt_restore_bwt_gpu.argtypes = [C.c_char_p]
    L.bwt_restore_sa_gpu.argtypes = [C.c_char_p, C.POINTER(BwtTGpu)]
    L.gpu_cpy_wrapper.restype = BwtTGpu
    L.gpu_cpy_wrapper.argtypes = [C.POINTER(BwtTGpu)]
    L.seed_gpu.restype = C.POINTER(MemSeedVGpu)
    L.seed_gpu.argtypes = [C.POINTER(GpuseedStorageVector)]
    L.free_gpuseed_data.argtypes = [C.POINTER(GpuseedStorageVector)]
    L.seed_gpu_last_n_reads.restype = C.c_uint64
    _LIB = L
    return L


def _err(L) -> str:
    return (L.bmh_last_error() or b"").decode()


def _np_ptr(a, t):
    return a.ctypes.data_as(t)


class Index:
    """FMD index resident in HBM (bmh_index_t)."""

    def __init__(self, handle, keep=None):
        self.handle = handle
        self._keep = keep

    @classmethod
    def upload(cls, idx, pac: np.ndarray | None = None, l_pac: int = 0) -> "Index":
        """idx: fmindex.FMDIndex on the host -> HBM (replaces gpu_cpy_wrapper)."""
        L = load_library()
        L2 = np.ascontiguousarray(idx.L2, dtype=np.uint64)
        bw = np.ascontiguousarray(idx.bwt_words, dtype=np.uint32)
        sa = np.ascontiguousarray(idx.sa, dtype=np.uint32)
        bits = np.ascontiguousarray(idx.sa_bits, dtype=np.uint32)
        h = L.bmh_index_upload(idx.primary, _np_ptr(L2, _u64p), idx.seq_len, _np_ptr(bw, _u32p), bw.shape[0],
                               idx.sa_intv, _np_ptr(sa, _u32p), idx.n_sa, _np_ptr(bits, _u32p),
                               _np_ptr(pac, _u8p) if pac is not None else None, l_pac)
        if not h:
            raise RuntimeError("bmh_index_upload: " + _err(L))
        return cls(h)

    @classmethod
    def from_device(cls, primary, L2, seq_len, bwt_t, sa_intv, sa_t, sa_bits_t, pac_t=None, l_pac=0) -> "Index":
        """Wrap torch tensors already in HBM (e.g. after the RCCL broadcast); tensors are kept alive."""
        L = load_library()
        L2a = np.ascontiguousarray(L2, dtype=np.uint64)
        h = L.bmh_index_from_device(primary, _np_ptr(L2a, _u64p), seq_len, bwt_t.data_ptr(), bwt_t.numel(), sa_intv,
                                    sa_t.data_ptr(), sa_t.numel(), sa_bits_t.data_ptr(),
                                    pac_t.data_ptr() if pac_t is not None else None, l_pac)
        if not h:
            raise RuntimeError("bmh_index_from_device: " + _err(L))
        return cls(h, keep=(bwt_t, sa_t, sa_bits_t, pac_t))

    def densify_sa(self, new_intv: int) -> None:
        """bmh_index_densify_sa: suffix-array samples of every new_intv-th row, computed on the device from the existing ones"""
        L = load_library()
        rc = L.bmh_index_densify_sa(self.handle, int(new_intv))
        if rc != 0:
            raise RuntimeError(f"bmh_index_densify_sa rc={rc}: " + _err(L))

    def probe(self, rows, what: str) -> np.ndarray:
        """bmh_index_probe: 'occ4' -> [n, 4] Occ counts, 'lf' -> LF(row), 'sa' -> SA[row] at the given rows (u64)"""
        import torch
        L = load_library()
        r = torch.from_numpy(np.ascontiguousarray(rows, dtype=np.uint64).view(np.int64)).cuda()
        code = {"occ4": 0, "lf": 1, "sa": 2}[what]
        out = torch.empty(r.numel() * (4 if code == 0 else 1), dtype=torch.int64, device="cuda")
        rc = L.bmh_index_probe(self.handle, r.data_ptr(), r.numel(), code, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise RuntimeError(f"bmh_index_probe rc={rc}: " + _err(L))
        torch.cuda.synchronize()
        o = out.cpu().numpy().view(np.uint64)
        return o.reshape(-1, 4) if code == 0 else o

    def free(self):
        if self.handle:
            load_library().bmh_index_free(self.handle)
            self.handle = None


class SeedWorkspace:
    def __init__(self, max_reads: int, max_bases: int, max_cands: int = 0, max_occ: int = 0):
        L = load_library()
        self.handle = L.bmh_seed_ws_create(max_reads, max_bases, max_cands, max_occ)
        if not self.handle:
            raise RuntimeError("bmh_seed_ws_create: " + _err(L))

    def seed_batch(self, index: Index, reads_t, offs_t, lens_t, min_seed_len: int = 19, stream: int = 0) -> SeedsT:
        """reads_t/offs_t/lens_t: torch CUDA tensors (uint8 ASCII, int32/uint32 offsets and lengths)."""
        L = load_library()
        out = SeedsT()
        rc = L.bmh_seed_batch(self.handle, index.handle, reads_t.data_ptr(), offs_t.data_ptr(), lens_t.data_ptr(),
                              lens_t.numel(), min_seed_len, stream, C.byref(out))
        if rc != 0:
            raise RuntimeError(f"bmh_seed_batch rc={rc}: " + _err(L))
        return out

    def timing(self):
        ms = (C.c_float * 7)()
        load_library().bmh_seed_last_timing(self.handle, ms)
        names = ("pack", "smem", "sort", "gather_scan", "expand", "locate", "total") if os.environ.get("BMH_SEED_FUSED") else ("pack", "forward", "backward", "filter_scan", "expand", "locate", "total")
        return dict(zip(names, list(ms)))

    def free(self):
        if self.handle:
            load_library().bmh_seed_ws_free(self.handle)
            self.handle = None


class ChainWorkspace:
    """Device job builder (bmh_chain_batch / bmh_chain_merge): seeds in HBM -> extension jobs in HBM -> regions."""

    def __init__(self, max_reads: int, max_seeds: int, opt: "ChainOpt | None" = None):
        L = load_library()
        self.handle = L.bmh_chain_ws_create(max_reads, max_seeds)
        if not self.handle:
            raise RuntimeError("bmh_chain_ws_create: " + _err(L))
        self.opt = opt or ChainOpt()
        if opt is None:
            L.bmh_chain_opt_default(C.byref(self.opt))

    def set_contigs(self, contigs) -> None:
        """contigs: list of (name, length) of the packed reference's sequences (bmh_chain_set_contigs)"""
        L = load_library()
        ln = np.ascontiguousarray([c[1] for c in contigs], dtype=np.int32)
        off = np.ascontiguousarray(np.concatenate([[0], np.cumsum(ln)[:-1]]), dtype=np.int64)
        rc = L.bmh_chain_set_contigs(self.handle, len(contigs), off.ctypes.data_as(C.c_void_p), ln.ctypes.data_as(C.c_void_p))
        if rc != 0:
            raise RuntimeError("bmh_chain_set_contigs: " + _err(L))

    def set_alt(self, is_alt) -> None:
        """is_alt: one flag per sequence of set_contigs' table (bmh_chain_set_alt; the .alt file of the index)"""
        L = load_library()
        a = np.ascontiguousarray(is_alt, dtype=np.uint8)
        rc = L.bmh_chain_set_alt(self.handle, len(a), a.ctypes.data_as(C.c_void_p))
        if rc != 0:
            raise RuntimeError("bmh_chain_set_alt: " + _err(L))

    def chain_batch(self, index: Index, reads_t, offs_t, lens_t, seeds: SeedsT, stream: int = 0) -> DevJobsT:
        L = load_library()
        out = DevJobsT()
        rc = L.bmh_chain_batch(self.handle, C.byref(self.opt), index.handle, reads_t.data_ptr(), offs_t.data_ptr(), lens_t.data_ptr(),
                               lens_t.numel(), C.byref(seeds), stream, C.byref(out))
        if rc != 0:
            raise RuntimeError(f"bmh_chain_batch rc={rc}: " + _err(L))
        self.last_jobs = out
        return out

    def set_materialize(self, on: bool) -> None:
        load_library().bmh_chain_set_materialize(self.handle, 1 if on else 0)

    def extend(self, out3_t, params: "ExtParams | None" = None, raw_t=None, stream: int = 0) -> None:
        """bmh_chain_extend: extension of the last chain_batch's jobs straight from their descriptors."""
        L = load_library()
        p = params or ExtParams.default()
        rc = L.bmh_chain_extend(self.handle, C.byref(p), out3_t.data_ptr(), raw_t.data_ptr() if raw_t is not None else None, stream)
        if rc != 0:
            raise RuntimeError(f"bmh_chain_extend rc={rc}: " + _err(L))

    def extend_merge(self, index: Index, reads_t, offs_t, lens_t, seeds: SeedsT, regs_t, params: "ExtParams | None" = None, stream: int = 0) -> DevJobsT:
        """bmh_chain_extend_merge: chaining, extension and region merge of one batch in one call (regs_t int32 [cap, 8], read order)"""
        L = load_library()
        out = DevJobsT()
        p = params or ExtParams.default()
        rc = L.bmh_chain_extend_merge(self.handle, C.byref(self.opt), index.handle, reads_t.data_ptr(), offs_t.data_ptr(), lens_t.data_ptr(),
                                      lens_t.numel(), C.byref(seeds), C.byref(p), regs_t.data_ptr(), int(regs_t.shape[0]), stream, C.byref(out))
        if rc != 0:
            raise RuntimeError(f"bmh_chain_extend_merge rc={rc}: " + _err(L))
        self.last_jobs = out
        return out

    def extend_merge_timing(self):
        L = load_library()
        ms = (C.c_float * 3)(); jobs = (C.c_uint64 * 2)()
        L.bmh_chain_extend_merge_timing(self.handle, ms, jobs)
        return {"extend_a": ms[0], "extend_b": ms[1], "stage": ms[2], "jobs_a": int(jobs[0]), "jobs_b": int(jobs[1])}

    def timing(self):
        ms = (C.c_float * 8)()
        load_library().bmh_chain_last_timing(self.handle, ms)
        return {"classify": ms[0], "lane": ms[1], "wave": ms[2], "to_counts": ms[3], "reads_small_scratch": int(ms[6]), "reads_large_scratch": int(ms[7])}

    def merge(self, out3_t, regs_t, stream: int = 0) -> None:
        L = load_library()
        rc = L.bmh_chain_merge(self.handle, out3_t.data_ptr(), regs_t.data_ptr(), stream)
        if rc != 0:
            raise RuntimeError(f"bmh_chain_merge rc={rc}: " + _err(L))

    def free(self):
        if self.handle:
            load_library().bmh_chain_ws_free(self.handle)
            self.handle = None


def cigar_batch(index: Index, reads_t, offs_t, lens_t, regs_t, n: int, sel_t=None, params: "ExtParams | None" = None, opt_w: int = 300,
                max_cigar: int = 32, md_cap: int = 128, stream: int = 0):
    """bmh_cigar_batch on torch CUDA tensors; returns (cigar uint32 [n, max_cigar], aln int32 [n, 8], md uint8 [n, md_cap]) tensors."""
    import torch
    L = load_library()
    p = params or ExtParams.default()
    dev = regs_t.device
    cigar = torch.zeros(max(n, 1), max_cigar, dtype=torch.int32, device=dev)
    aln = torch.zeros(max(n, 1), 8, dtype=torch.int32, device=dev)
    md = torch.zeros(max(n, 1), md_cap, dtype=torch.uint8, device=dev)
    rc = L.bmh_cigar_batch(index.handle, reads_t.data_ptr(), offs_t.data_ptr(), lens_t.data_ptr(), regs_t.data_ptr(), int(regs_t.shape[1]),
                           sel_t.data_ptr() if sel_t is not None else None, n, C.byref(p), opt_w, max_cigar, cigar.data_ptr(), aln.data_ptr(),
                           md_cap, md.data_ptr(), stream)
    if rc != 0:
        raise RuntimeError(f"bmh_cigar_batch rc={rc}: " + _err(L))
    return cigar, aln, md


def sam_select_device(po: "PostOpt", fin_t, fin_per_read_t, h_rec_t=None, stream: int = 0):
    """bmh_sam_select_device on torch CUDA tensors (fin int32 [m, 16], fin_per_read int32 [n_reads], h_rec int32 [n_reads] for pairs):
    returns (sel int32 [n_sel], slot int32 [m]) on the device."""
    import torch
    L = load_library()
    m, n_reads = int(fin_t.shape[0]), int(fin_per_read_t.shape[0])
    dev = fin_t.device
    sel = torch.empty(max(m, 1), dtype=torch.int32, device=dev); slot = torch.empty(max(m, 1), dtype=torch.int32, device=dev)
    wb = int(L.bmh_sam_select_work(n_reads, m))
    work = torch.empty(wb, dtype=torch.uint8, device=dev)
    k = L.bmh_sam_select_device(C.byref(po), fin_t.data_ptr(), fin_per_read_t.data_ptr(), h_rec_t.data_ptr() if h_rec_t is not None else None, n_reads, m,
                                sel.data_ptr(), slot.data_ptr(), work.data_ptr(), wb, stream)
    if k < 0:
        raise RuntimeError(f"bmh_sam_select_device rc={k}: " + _err(L))
    return sel[:k], slot[:m]


def cigar_pack(aln_t, cigar_t, md_t=None, stream: int = 0):
    """bmh_cigar_pack_sizes + bmh_cigar_pack on the outputs of cigar_batch: returns (off int32 [n + 1], packed int32 [words]) on the device"""
    import torch
    L = load_library()
    n = int(aln_t.shape[0])
    dev = aln_t.device
    off = torch.empty(n + 2, dtype=torch.int32, device=dev)
    wb = int(L.bmh_cigar_pack_work(n))
    work = torch.empty(wb, dtype=torch.uint8, device=dev)
    words = L.bmh_cigar_pack_sizes(aln_t.data_ptr(), n, 1 if md_t is not None else 0, off.data_ptr(), work.data_ptr(), wb, stream)
    if words < 0:
        raise RuntimeError(f"bmh_cigar_pack_sizes rc={words}: " + _err(L))
    packed = torch.empty(max(int(words), 1), dtype=torch.int32, device=dev)
    rc = L.bmh_cigar_pack(aln_t.data_ptr(), cigar_t.data_ptr(), int(cigar_t.shape[1]), md_t.data_ptr() if md_t is not None else None,
                          int(md_t.shape[1]) if md_t is not None else 0, n, off.data_ptr(), packed.data_ptr(), stream)
    if rc != 0:
        raise RuntimeError(f"bmh_cigar_pack rc={rc}: " + _err(L))
    torch.cuda.synchronize()
    return off[:n + 1], packed[:int(words)]


def sam_text_device(po: "PostOpt", names, reads_t, offs_t, lens_t, contigs, fin_t, fin_per_read_t, slot_t, aln_t, cig_off_t, packed_t, h_rec_t=None, unflag_t=None,
                    stream: int = 0) -> bytes:
    """bmh_sam_text_sizes + bmh_sam_text_write on torch CUDA tensors: the SAM text of a batch written on the device.  names: list of str;
    contigs: list of (name, length); reads_t / offs_t / lens_t: the batch's ASCII reads; the rest as sam_select_device / cigar_batch / cigar_pack
    left them.  Returns the text as bytes."""
    import torch
    L = load_library()
    dev = fin_t.device
    n = int(fin_per_read_t.shape[0])
    enc = [x.encode() + b"\0" for x in names]
    nblob = torch.from_numpy(np.frombuffer(b"".join(enc) or b"\0", dtype=np.uint8).copy()).to(dev)
    noff = torch.from_numpy(np.concatenate([[0], np.cumsum([len(e) for e in enc])]).astype(np.int64)).to(dev)            # [n + 1]
    cenc = [c[0].encode() + b"\0" for c in contigs]
    cblob = torch.from_numpy(np.frombuffer(b"".join(cenc), dtype=np.uint8).copy()).to(dev)
    cnoff = torch.from_numpy(np.concatenate([[0], np.cumsum([len(e) for e in cenc])]).astype(np.int32)).to(dev)         # [n_contigs + 1]
    coff = torch.from_numpy(np.concatenate([[0], np.cumsum([c[1] for c in contigs])[:-1]]).astype(np.int64)).to(dev)
    d = SamDev(n, nblob.data_ptr(), noff.data_ptr(), reads_t.data_ptr(), offs_t.data_ptr(), lens_t.data_ptr(), len(contigs), cblob.data_ptr(), cnoff.data_ptr(), coff.data_ptr(),
               fin_t.data_ptr(), fin_per_read_t.data_ptr(), slot_t.data_ptr(), aln_t.data_ptr(), cig_off_t.data_ptr(), packed_t.data_ptr(),
               h_rec_t.data_ptr() if h_rec_t is not None else None, unflag_t.data_ptr() if unflag_t is not None else None)
    wb = int(L.bmh_sam_text_work(n))
    work = torch.empty(wb, dtype=torch.uint8, device=dev)
    toff = torch.empty(n + 2, dtype=torch.int64, device=dev)
    total = L.bmh_sam_text_sizes(C.byref(po), C.byref(d), toff.data_ptr(), work.data_ptr(), wb, stream)
    if total < 0:
        raise RuntimeError(f"bmh_sam_text_sizes rc={total}: " + _err(L))
    text = torch.empty(max(int(total), 1), dtype=torch.uint8, device=dev)
    rc = L.bmh_sam_text_write(C.byref(po), C.byref(d), toff.data_ptr(), text.data_ptr(), work.data_ptr(), wb, stream)
    if rc == 0:
        rc = L.bmh_sam_text_check(work.data_ptr(), n, stream)
    if rc != 0:
        raise RuntimeError(f"bmh_sam_text_write rc={rc}: " + _err(L))
    return text[: int(total)].cpu().numpy().tobytes()


class CapacityError(RuntimeError):
    """a BMH_ECAPACITY return: the device form of a stage met a read beyond its fixed limits; the caller may take the host form"""


def finalize_regs_device(index: "Index", copt, ep, po, reads_t, offs_t, regs_t, n_regs: int, regs_per_read_ptr: int, frac_rep_ptr: int, n_reads: int,
                         contigs=None, out_t=None, opr_t=None, stream: int = 0):
    """bmh_finalize_regs_device on torch CUDA tensors / device pointers (regs_per_read_ptr, frac_rep_ptr: bmh_dev_jobs_t.d_regs_per_read /
    d_frac_rep).  Returns (out int32 [m, 16] view, out_per_read int32 [n_reads]) on the device."""
    import torch
    L = load_library()
    dev = regs_t.device
    if out_t is None:
        out_t = torch.empty(max(n_regs, 1), 16, dtype=torch.int32, device=dev)
    if opr_t is None:
        opr_t = torch.empty(max(n_reads, 1), dtype=torch.int32, device=dev)
    off = np.ascontiguousarray(np.concatenate([[0], np.cumsum([c[1] for c in contigs])[:-1]]), dtype=np.int64) if contigs and len(contigs) > 1 else None
    m = L.bmh_finalize_regs_device(index.handle, C.byref(copt), C.byref(ep), C.byref(po), reads_t.data_ptr(), offs_t.data_ptr(), n_reads,
                                   regs_t.data_ptr(), n_regs, regs_per_read_ptr, frac_rep_ptr, len(contigs) if off is not None else 1,
                                   off.ctypes.data_as(C.c_void_p) if off is not None else None, out_t.data_ptr(), opr_t.data_ptr(), stream)
    if m == -3:
        raise CapacityError(f"bmh_finalize_regs_device rc={m}: " + _err(L))
    if m < 0:
        raise RuntimeError(f"bmh_finalize_regs_device rc={m}: " + _err(L))
    return out_t[:m], opr_t


def finalize_pairs(copt, ep, po, genome_len: int, pac: np.ndarray, reads_flat: np.ndarray, read_offs: np.ndarray, read_lens: np.ndarray,
                   regs: np.ndarray, regs_per_read: np.ndarray, frac_rep: np.ndarray, contigs=None, n_threads: int = 1, pe=None, out=None, device=None):
    """bmh_finalize_pairs -> (fin [m,16], per_read, h_rec, unflag, pes [4,5]).  out: optional preallocated int32 [cap,16] buffer
    (a caller that runs batch after batch keeps one, so the pages are not faulted in on every call).
    device = (Index, reads_t ASCII, offs_t, stream): bmh_finalize_pairs_dev -- the mate rescue's local alignments as one batch on the device"""
    L = load_library()
    if pe is None:
        pe = PeOpt(); L.bmh_pe_opt_default(C.byref(pe))
    n = len(read_lens)
    a = lambda x, dt: np.ascontiguousarray(x, dtype=dt)
    regs = a(regs, np.int32)
    cap = len(regs) + 2 * n + 1024          # mate rescue adds a few regions per pair; BMH_ECAPACITY (-3) if this is short: retried larger
    opr = np.zeros(max(n, 1), np.uint32); h = np.zeros(max(n, 1), np.int32); uf = np.zeros(max(n, 1), np.int32)
    pes = np.zeros((4, 5), np.float64)
    ln = a([c[1] for c in contigs], np.int32) if contigs else None
    off = a(np.concatenate([[0], np.cumsum(ln)[:-1]]), np.int64) if contigs else None
    keep = [a(pac, np.uint8), a(reads_flat, np.uint8), a(read_offs, np.uint64), a(read_lens, np.uint32), a(regs_per_read, np.uint32), a(frac_rep, np.float32)]
    if out is not None and out.dtype == np.int32 and out.ndim == 2 and out.shape[1] == 16 and out.flags.c_contiguous and len(out) >= cap:
        cap = len(out)
    else:
        out = None
    for attempt in range(4):
        if out is None or len(out) < cap:
            out = np.empty((cap, 16), np.int32)
        fn, head = L.bmh_finalize_pairs, ()
        if device is not None:
            fn, head = L.bmh_finalize_pairs_dev, (device[0].handle, device[1].data_ptr(), device[2].data_ptr(), device[3] if len(device) > 3 else None)
        m = fn(*head, C.byref(copt), C.byref(ep), C.byref(po), C.byref(pe), genome_len, _np_ptr(keep[0], _u8p), n, _np_ptr(keep[1], _u8p),
                                 _np_ptr(keep[2], _u64p), _np_ptr(keep[3], _u32p), _np_ptr(regs, _i32p), _np_ptr(keep[4], _u32p),
                                 keep[5].ctypes.data_as(C.POINTER(C.c_float)), len(contigs) if contigs else 1,
                                 off.ctypes.data_as(C.c_void_p) if contigs else None, ln.ctypes.data_as(C.c_void_p) if contigs else None,
                                 _np_ptr(out, _i32p), cap, _np_ptr(opr, _u32p), _np_ptr(h, _i32p), _np_ptr(uf, _i32p), pes.ctypes.data_as(C.c_void_p), n_threads)
        if m != -3:
            break
        cap = cap * 3 + 50 * n               # at most max_matesw rescued regions per read
    if m < 0:
        raise RuntimeError("bmh_finalize_pairs: " + _err(L))
    return out[:m], opr[:n], h[:n], uf[:n], pes


def format_sam(po: "PostOpt", names, reads_flat: np.ndarray, read_offs: np.ndarray, read_lens: np.ndarray, contigs, fin: np.ndarray,
               fin_per_read: np.ndarray, slot: np.ndarray, aln: np.ndarray, cigar: np.ndarray, md: np.ndarray, h_rec=None, unflag=None,
               as_bytes: bool = False):
    """bmh_format_sam (or bmh_format_sam_pe when h_rec / unflag are given) on numpy arrays; contigs = list of (name, length).
    names: list of str, or (blob uint8 array of NUL-terminated names, uint64 offsets).  Returns the text as bytes if
    as_bytes, as a uint8 array over the library's own buffer if as_bytes == "view", else as str."""
    L = load_library()
    if isinstance(names, tuple):
        nblob, noff = np.ascontiguousarray(names[0], dtype=np.uint8), np.ascontiguousarray(names[1], dtype=np.uint64)
    else:
        enc = [n.encode() + b"\0" for n in names]
        nblob = np.frombuffer(b"".join(enc), dtype=np.uint8)
        noff = np.concatenate([[0], np.cumsum([len(e) for e in enc])[:-1]]).astype(np.uint64) if enc else np.zeros(1, np.uint64)
    n_names = len(noff) if len(noff) else 0
    cn = (C.c_char_p * len(contigs))(*[c[0].encode() for c in contigs])
    off = np.ascontiguousarray(np.concatenate([[0], np.cumsum([c[1] for c in contigs])[:-1]]), dtype=np.int64)
    a = lambda x, dt: np.ascontiguousarray(x, dtype=dt)
    keep = [a(reads_flat, np.uint8), a(read_offs, np.uint64), a(read_lens, np.uint32), a(fin, np.int32), a(fin_per_read, np.uint32), a(slot, np.int64),
            a(aln, np.int32), a(cigar, np.uint32), a(md, np.uint8)]
    ln = C.c_size_t()
    if h_rec is not None:
        hh, uu = a(h_rec, np.int32), a(unflag, np.int32)
        p = L.bmh_format_sam_pe(C.byref(po), len(read_lens), nblob.ctypes.data_as(C.c_void_p), _np_ptr(noff, _u64p), _np_ptr(keep[0], _u8p), _np_ptr(keep[1], _u64p), _np_ptr(keep[2], _u32p), len(contigs), cn,
                                off.ctypes.data_as(C.c_void_p), _np_ptr(keep[3], _i32p), _np_ptr(keep[4], _u32p), _np_ptr(hh, _i32p), _np_ptr(uu, _i32p),
                                keep[5].ctypes.data_as(C.c_void_p), _np_ptr(keep[6], _i32p), _np_ptr(keep[7], _u32p), int(keep[7].shape[1]),
                                keep[8].ctypes.data_as(C.c_void_p), int(keep[8].shape[1]), C.byref(ln))
    else:
        p = L.bmh_format_sam(C.byref(po), len(read_lens), nblob.ctypes.data_as(C.c_void_p), _np_ptr(noff, _u64p), _np_ptr(keep[0], _u8p), _np_ptr(keep[1], _u64p), _np_ptr(keep[2], _u32p), len(contigs), cn,
                             off.ctypes.data_as(C.c_void_p), _np_ptr(keep[3], _i32p), _np_ptr(keep[4], _u32p), keep[5].ctypes.data_as(C.c_void_p),
                             _np_ptr(keep[6], _i32p), _np_ptr(keep[7], _u32p), int(keep[7].shape[1]), keep[8].ctypes.data_as(C.c_void_p), int(keep[8].shape[1]),
                             C.byref(ln))
    if not p:
        raise RuntimeError("bmh_format_sam: " + _err(L))
    if as_bytes == "view":                  # the library's buffer itself as a uint8 array (freed with the array): no copy of the text
        import weakref
        arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(max(int(ln.value), 1),))[: int(ln.value)]
        weakref.finalize(arr.base if arr.base is not None else arr, L.bmh_free, C.c_void_p(p))
        return arr
    raw = C.string_at(p, ln.value)
    L.bmh_free(p)
    return raw if as_bytes else raw.decode()


def dev_jobs_to_host(j: DevJobsT, n_reads: int) -> dict:
    """Copy a bmh_dev_jobs_t to numpy arrays (test helper)."""
    import torch

    def rd(ptr, n, dt, tdt):
        if n == 0 or not ptr:
            return np.zeros(0, dt)
        buf = torch.empty(n, dtype=tdt, device="cuda")
        _memcpy_d2d(buf.data_ptr(), ptr, n * buf.element_size())
        return buf.cpu().numpy().view(dt)
    nj = int(j.n_jobs)
    out = {k: rd(getattr(j, "d_" + k), nj, np.uint32, torch.int32) for k in ("qoff", "qlen", "toff", "tlen", "h0", "job_read", "job_reg", "job_side")}
    out["q"] = rd(j.d_q, int(j.q_bytes), np.uint8, torch.uint8)
    out["t"] = rd(j.d_t, int(j.t_bytes), np.uint8, torch.uint8)
    out["regs_per_read"] = rd(j.d_regs_per_read, n_reads, np.uint32, torch.int32)
    out["frac_rep"] = rd(j.d_frac_rep, n_reads, np.float32, torch.float32)
    return out


def seeds_to_host(s: SeedsT, n_reads: int) -> dict:
    """Copy a bmh_seeds_t to numpy arrays (test/bench helper; uses torch only to read HBM)."""
    import torch

    def rd(ptr, n, dt, tdt):
        if n == 0:
            return np.zeros(0, dt)
        buf = torch.empty(n, dtype=tdt, device="cuda")
        _memcpy_d2d(buf.data_ptr(), ptr, n * buf.element_size())
        return buf.cpu().numpy().view(dt)
    ns = int(s.n_seeds)
    return dict(rbeg=rd(s.d_rbeg, ns, np.uint64, torch.int64), qbeg=rd(s.d_qbeg, 2 * ns, np.int32, torch.int32).reshape(-1, 2),
                score=rd(s.d_score, ns, np.uint32, torch.int32), n_ref_pos=rd(s.d_n_ref_pos, n_reads, np.uint32, torch.int32),
                prefix=rd(s.d_prefix, n_reads, np.uint32, torch.int32))


def _memcpy_d2d(dst: int, src: int, nbytes: int) -> None:
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    rc = hip.hipMemcpy(dst, src, nbytes, 3)  # hipMemcpyDeviceToDevice
    if rc != 0:
        raise RuntimeError(f"hipMemcpy d2d failed rc={rc}")


def extend_batch(q_t, qoff_t, qlen_t, t_t, toff_t, tlen_t, h0_t, out_t, params: ExtParams | None = None, raw_t=None,
                 stream: int = 0) -> None:
    """All arguments torch CUDA tensors; out_t int32 [n,3]; raw_t int32 [n,6] or None.  Asynchronous."""
    L = load_library()
    p = params or ExtParams.default()
    rc = L.bmh_extend_batch(q_t.data_ptr(), qoff_t.data_ptr(), qlen_t.data_ptr(), t_t.data_ptr(), toff_t.data_ptr(),
                            tlen_t.data_ptr(), h0_t.data_ptr(), qlen_t.numel(), C.byref(p), out_t.data_ptr(),
                            raw_t.data_ptr() if raw_t is not None else None, stream)
    if rc != 0:
        raise RuntimeError(f"bmh_extend_batch rc={rc}: " + _err(L))


def seed_file(index_prefix: str, read_file: str, min_seed_len: int = 19) -> dict:
    """The reference's call sequence (src/fastmap.c:436-465) through the drop-in C ABI:
    bwt_restore_bwt_gpu -> bwt_restore_sa_gpu -> gpu_cpy_wrapper -> seed_gpu -> free_gpuseed_data."""
    L = load_library()
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    d = GpuseedStorageVector()
    d.query_file = index_prefix.encode()
    d.read_file = read_file.encode()
    d.file_bytes_skip = 0
    d.min_seed_size = min_seed_len
    d.is_smem = 1
    d.bwt = L.bwt_restore_bwt_gpu((index_prefix + ".bwt").encode())
    L.bwt_restore_sa_gpu((index_prefix + ".sa").encode(), d.bwt)
    d.bwt_gpu = L.gpu_cpy_wrapper(d.bwt)
    d.pre_calc_seed_len = 13
    d.pre_calc_seed_intervals_flag = 0
    res = L.seed_gpu(C.byref(d))
    n_reads = int(L.seed_gpu_last_n_reads())
    L.free_gpuseed_data(C.byref(d))
    r = res.contents
    n_ref = np.ctypeslib.as_array(r.n_ref_pos_fow_rev_results, shape=(max(n_reads, 1),))[:n_reads].copy()
    prefix = np.ctypeslib.as_array(r.n_ref_pos_fow_rev_prefix_sums, shape=(max(n_reads, 1),))[:n_reads].copy()
    ns = int(n_ref.sum())
    out = dict(rbeg=np.ctypeslib.as_array(r.rbeg, shape=(max(ns, 1),))[:ns].copy(),
               qbeg=np.ctypeslib.as_array(r.qbeg, shape=(max(2 * ns, 1),))[:2 * ns].copy().reshape(-1, 2),
               score=np.ctypeslib.as_array(r.score, shape=(max(ns, 1),))[:ns].copy(),
               n_ref_pos=n_ref, prefix=prefix, file_bytes=int(r.file_bytes_skip))
    for p in (r.rbeg, r.qbeg, r.score, r.n_ref_pos_fow_rev_results, r.n_ref_pos_fow_rev_prefix_sums):
        libc.free(C.cast(p, C.c_void_p))
    libc.free(C.cast(res, C.c_void_p))
    return out


class HostJobs:
    """Host job builder (bmh_build_jobs): chains -> filtered chains -> extension jobs, per read."""

    def __init__(self, genome_fwd: np.ndarray, reads: np.ndarray, read_offs: np.ndarray, read_lens: np.ndarray, seeds: dict,
                 n_threads: int = 0, opt: "ChainOpt | None" = None, contigs=None, pac: "np.ndarray | None" = None):
        """contigs: optional list of (name, length) -- the sequences of the packed reference, in order; pac: the 2-bit forward strand
        if the caller has it already (packing a 3.1 Gbp genome takes seconds; self.t_pack tells how long it took here)"""
        import time as _time
        L = load_library()
        self.L = L
        o = opt or ChainOpt()
        if opt is None:
            L.bmh_chain_opt_default(C.byref(o))
        self.opt = o
        self._genome = genome_fwd
        l_pac = int(genome_fwd) if np.isscalar(genome_fwd) else int(genome_fwd.shape[0])       # (a length, when the caller passes the pac)
        self.l_pac = l_pac
        _t0 = _time.time()
        if pac is None:
            pad = (-l_pac) % 4
            codes = np.concatenate([genome_fwd, np.zeros(pad, np.uint8)]).reshape(-1, 4)
            pac = np.ascontiguousarray(((codes[:, 0] << 6) | (codes[:, 1] << 4) | (codes[:, 2] << 2) | codes[:, 3]).astype(np.uint8))
        else:
            pac = np.ascontiguousarray(pac, dtype=np.uint8)
        self.t_pack = _time.time() - _t0
        a = lambda x, dt: np.ascontiguousarray(x, dtype=dt)
        self._keep = [pac, a(reads, np.uint8), a(read_offs, np.uint64), a(read_lens, np.uint32), a(seeds["rbeg"], np.uint64),
                      a(seeds["qbeg"], np.int32), a(seeds["score"], np.uint32), a(seeds["n_ref_pos"], np.uint32), a(seeds["prefix"], np.uint32)]
        k = self._keep
        self.n_contigs = len(contigs) if contigs else 1
        self.ctg_len = np.ascontiguousarray([c[1] for c in contigs], dtype=np.int32) if contigs else None
        self.ctg_off = np.ascontiguousarray(np.concatenate([[0], np.cumsum(self.ctg_len)[:-1]]), dtype=np.int64) if contigs else None
        self.h = L.bmh_build_jobs(C.byref(o), l_pac, _np_ptr(k[0], _u8p), self.n_contigs,
                                  self.ctg_off.ctypes.data_as(C.c_void_p) if contigs else None, self.ctg_len.ctypes.data_as(C.c_void_p) if contigs else None,
                                  len(k[3]), _np_ptr(k[1], _u8p),
                                  _np_ptr(k[2], _u64p), _np_ptr(k[3], _u32p), _np_ptr(k[4], _u64p), _np_ptr(k[5], _i32p),
                                  _np_ptr(k[6], _u32p), _np_ptr(k[7], _u32p), _np_ptr(k[8], _u32p), n_threads or (os.cpu_count() or 1))
        if not self.h:
            raise RuntimeError("bmh_build_jobs: " + _err(L))
        sz = [C.c_uint64() for _ in range(4)]
        L.bmh_jobs_sizes(self.h, *[C.byref(x) for x in sz])
        self.n_jobs, self.n_regs, qb, tb = (int(x.value) for x in sz)
        ptrs = [C.c_void_p() for _ in range(11)]
        L.bmh_jobs_arrays(self.h, *[C.byref(p) for p in ptrs])

        def view(p, n, dt):
            if n == 0:
                return np.zeros(0, dt)
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8 if dt == np.uint8 else C.c_uint32)), shape=(n,)).view(dt)
        nj = self.n_jobs
        self.q = view(ptrs[0], qb, np.uint8); self.qoff = view(ptrs[1], nj, np.uint32); self.qlen = view(ptrs[2], nj, np.uint32)
        self.t = view(ptrs[3], tb, np.uint8); self.toff = view(ptrs[4], nj, np.uint32); self.tlen = view(ptrs[5], nj, np.uint32)
        self.h0 = view(ptrs[6], nj, np.uint32); self.job_read = view(ptrs[7], nj, np.uint32); self.job_reg = view(ptrs[8], nj, np.uint32)
        self.job_side = view(ptrs[9], nj, np.uint32); self.regs_per_read = view(ptrs[10], len(k[3]), np.uint32)

    def jobs(self):
        q = self.q if self.q.size else np.zeros(1, np.uint8)
        t = self.t if self.t.size else np.zeros(1, np.uint8)
        return q, self.qoff, self.qlen, t, self.toff, self.tlen, self.h0

    def merge(self, out3: np.ndarray) -> np.ndarray:
        out3 = np.ascontiguousarray(out3, dtype=np.int32)
        regs = np.zeros((max(self.n_regs, 1), 8), np.int32)
        self.L.bmh_merge_regs(self.h, _np_ptr(out3, _i32p), _np_ptr(regs, _i32p))
        return regs[: self.n_regs]

    def frac_rep(self) -> np.ndarray:
        n = len(self._keep[3])
        return np.ctypeslib.as_array(self.L.bmh_jobs_frac_rep(self.h), shape=(max(n, 1),))[:n].copy()

    def finalize(self, regs: np.ndarray, flag_all: bool = False, id0: int = 0, params: "ExtParams | None" = None, n_threads: int = 1):
        """bmh_finalize_regs on the regions of merge(): -> (out int32 [m, 16], out_per_read uint32 [n_reads])"""
        L = self.L
        po = PostOpt(); L.bmh_post_opt_default(C.byref(po)); po.flag_all = 1 if flag_all else 0; po.id0 = id0
        co = self.opt
        ep = params or ExtParams.default()
        k = self._keep
        regs = np.ascontiguousarray(regs, dtype=np.int32)
        out = np.zeros((max(len(regs), 1), 16), np.int32); opr = np.zeros(max(len(k[3]), 1), np.uint32)
        fr = np.ascontiguousarray(self.frac_rep(), dtype=np.float32)
        m = L.bmh_finalize_regs(C.byref(co), C.byref(ep), C.byref(po), self.l_pac, _np_ptr(k[0], _u8p), len(k[3]), _np_ptr(k[1], _u8p),
                                _np_ptr(k[2], _u64p), _np_ptr(regs, _i32p), _np_ptr(np.ascontiguousarray(self.regs_per_read), _u32p),
                                fr.ctypes.data_as(C.POINTER(C.c_float)), self.n_contigs,
                                self.ctg_off.ctypes.data_as(C.c_void_p) if self.ctg_off is not None else None,
                                _np_ptr(out, _i32p), _np_ptr(opr, _u32p), n_threads)
        if m < 0:
            raise RuntimeError("bmh_finalize_regs: " + _err(L))
        return out[:m], opr[: len(k[3])]

    def free(self):
        if self.h:
            self.L.bmh_jobs_free(self.h)
            self.h = None
