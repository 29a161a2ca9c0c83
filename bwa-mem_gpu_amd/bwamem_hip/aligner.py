"""Single-end `gase_aln` on the device-resident path: index files + FASTA reads -> SAM.

The reference's CLI keeps working on this library through the drop-in headers (INTEGRATION.md sections 1-2); this module
is the same job done through the device-level C ABI instead -- seeding, chaining / job construction, extension and the
region merge on the GPU (bmh_seed_batch, bmh_chain_batch, bmh_chain_extend, bmh_chain_merge), the region tail on host
threads as in the reference (bmh_finalize_regs), CIGAR / NM / MD on the GPU (bmh_cigar_batch), SAM text on the host
(bmh_format_sam).  It writes the records the reference writes, byte for byte (tests/test_gpu_parity.py).  torch is used
for device memory only.  Interleaved pairs (gase_aln -p) go through bmh_finalize_pairs / bmh_format_sam_pe; their
insert-size statistics are per batch as in the reference, so identical output needs the reference's batching (one batch
here = batch_reads reads).  ALT contigs come from <prefix>.alt as in the reference; read groups and FASTQ qualities are not handled.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import fmindex
from .lib import (ChainOpt, ChainWorkspace, ExtParams, Index, PeOpt, PostOpt, SeedWorkspace, _memcpy_d2d, _np_ptr, _i32p, _u32p, _u64p, _u8p,
                  cigar_batch, finalize_pairs, format_sam, load_library)

_NT4 = np.full(256, 4, np.uint8)
for _i, _c in enumerate(b"ACGT"):
    _NT4[_c] = _i
    _NT4[_c + 32] = _i


def read_ann(prefix: str):
    """contigs [(name, length)] and l_pac from <prefix>.ann (bns_restore, src/bntseq.c:98-160)"""
    with open(prefix + ".ann") as f:
        l_pac, n_seqs, _ = f.readline().split()
        contigs = []
        for _ in range(int(n_seqs)):
            name = f.readline().split()[1]
            _, ln, _ = f.readline().split()
            contigs.append((name, int(ln)))
    return contigs, int(l_pac)


def read_alt(prefix: str, contigs) -> np.ndarray:
    """is_alt per sequence from <prefix>.alt, the reference's way (bns_restore, src/bntseq.c:179-200): the first field of every line
    that does not start with '@' names an ALT contig; names the index does not hold are ignored.  No file: no ALT contigs."""
    alt = np.zeros(len(contigs), np.uint8)
    path = prefix + ".alt"
    if not os.path.exists(path):
        return alt
    where = {}
    for i, c in enumerate(contigs):
        where[c[0]] = i                                   # (kh_put keeps one entry per name; a later duplicate overwrites its value)
    import re
    with open(path, "rb") as f:
        lines = f.read().split(b"\n")
    for k, line in enumerate(lines):
        parts = re.split(rb"[\t\r]", line, maxsplit=1)
        if k == len(lines) - 1 and len(parts) == 1:
            break                                         # (the reference completes a name at a tab, CR or LF: a bare name before EOF is never looked up)
        tok = parts[0].decode("latin-1")
        if tok and tok[0] != "@" and tok in where:
            alt[where[tok]] = 1
    return alt


class ReadSet:
    """reads of a FASTA file as flat arrays: ascii bases back to back + offsets / lengths, names as a NUL-separated blob"""

    def __init__(self, ascii_, offs, lens, name_blob, name_off, codes=None):
        self.ascii, self.offs, self.lens, self.name_blob, self.name_off = ascii_, offs, lens, name_blob, name_off
        self.codes = codes                      # nt4 codes of the letters when the loader made them (bmh_reads_load_fasta), else None

    def __len__(self):
        return len(self.lens)

    def slice(self, b0: int, b1: int) -> "ReadSet":
        b1 = min(b1, len(self))
        a0 = int(self.offs[b0]); a1 = int(self.offs[b1 - 1] + self.lens[b1 - 1])
        n0 = int(self.name_off[b0]); n1 = int(self.name_off[b1]) if b1 < len(self) else len(self.name_blob)
        return ReadSet(self.ascii[a0:a1], self.offs[b0:b1] - np.uint64(a0), self.lens[b0:b1], self.name_blob[n0:n1], self.name_off[b0:b1] - np.uint64(n0),
                       codes=None if self.codes is None else self.codes[a0:a1])

    @classmethod
    def from_lists(cls, names, seqs) -> "ReadSet":
        seqs = [np.frombuffer(s.encode(), dtype=np.uint8) if isinstance(s, str) else np.asarray(s, dtype=np.uint8) for s in seqs]
        lens = np.array([len(s) for s in seqs], np.uint32)
        offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64) if len(seqs) else np.zeros(0, np.uint64)
        ascii_ = np.concatenate(seqs) if len(seqs) and lens.sum() else np.zeros(1, np.uint8)
        enc = [n.encode() + b"\0" for n in names]
        blob = np.frombuffer(b"".join(enc), dtype=np.uint8) if enc else np.zeros(1, np.uint8)
        noff = np.concatenate([[0], np.cumsum([len(e) for e in enc])[:-1]]).astype(np.uint64) if enc else np.zeros(0, np.uint64)
        return cls(ascii_, offs, lens, blob, noff)


def read_fasta_reads(path: str) -> ReadSet:
    """one '>' header line and one sequence line per read (the only layout the reference's seeding library parses,
    src/GPUSeed/seed_gen.cu:1698-1728), through the library's loader (bmh_reads_load_fasta: host threads, letters + nt4 codes)"""
    from .lib import load_fasta_reads
    d = load_fasta_reads(path)
    if len(d["lens"]) == 0:
        return ReadSet.from_lists([], [])
    return ReadSet(d["ascii"], d["offs"], d["lens"], d["names"], d["name_offs"], codes=d["codes"])


def read_fasta_reads_numpy(path: str) -> ReadSet:
    """the same parse with numpy array operations (what read_fasta_reads was before the library had a loader; kept as its cross-check)"""
    buf = np.fromfile(path, dtype=np.uint8)
    if buf.size == 0:
        return ReadSet.from_lists([], [])
    if buf[-1] != 10:
        buf = np.concatenate([buf, np.array([10], np.uint8)])
    nl = np.flatnonzero(buf == 10)
    starts = np.concatenate([[0], nl[:-1] + 1]); ends = nl.copy()
    cr = (ends > starts) & (buf[np.maximum(ends - 1, 0)] == 13)
    ends = ends - cr
    keep = ends > starts
    starts, ends = starts[keep], ends[keep]
    is_hdr = buf[starts] == ord(">")
    if (~is_hdr).sum() != is_hdr.sum() or not is_hdr[0::2].all() or is_hdr[1::2].any():
        raise ValueError("reads file: expected alternating '>' header and sequence lines")
    hs, he = starts[0::2] + 1, ends[0::2]
    ss, se = starts[1::2], ends[1::2]
    lens = (se - ss).astype(np.uint32)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    # gather the sequence bytes (drop headers / newlines): a mask over the buffer
    mark = np.zeros(buf.size + 1, np.int8); np.add.at(mark, ss, 1); np.add.at(mark, se, -1)
    ascii_ = buf[np.cumsum(mark[:-1]) > 0]
    # names: header up to the first blank, NUL-terminated in place
    hb = buf.copy()
    blank = (hb == 32) | (hb == 9)
    first_blank = np.full(len(hs), -1, np.int64)
    bl = np.flatnonzero(blank)
    if bl.size:
        j = np.searchsorted(bl, hs)
        ok = (j < bl.size)
        cand = np.where(ok, bl[np.minimum(j, bl.size - 1)], -1)
        first_blank = np.where(ok & (cand < he), cand, -1)
    name_end = np.where(first_blank >= 0, first_blank, he)
    nlen = (name_end - hs).astype(np.int64)
    # a trailing "/<digit>" is not part of the name (trim_readno, src/bwa.c:27-31)
    e1, e2 = buf[np.maximum(name_end - 1, 0)], buf[np.maximum(name_end - 2, 0)]
    trim = (nlen > 2) & (e2 == ord("/")) & (e1 >= ord("0")) & (e1 <= ord("9"))
    name_end = name_end - 2 * trim
    nlen = (name_end - hs).astype(np.int64)
    noff = np.concatenate([[0], np.cumsum(nlen + 1)[:-1]]).astype(np.uint64)
    blob = np.zeros(int((nlen + 1).sum()), np.uint8)
    nmark = np.zeros(buf.size + 1, np.int8); np.add.at(nmark, hs, 1); np.add.at(nmark, name_end, -1)
    src = np.flatnonzero(np.cumsum(nmark[:-1]) > 0)
    dst = np.arange(src.size) + np.repeat(np.arange(len(hs)), nlen)          # one NUL after every name
    blob[dst] = buf[src]
    return ReadSet(ascii_ if ascii_.size else np.zeros(1, np.uint8), offs, lens, blob, noff)


class Aligner:
    def __init__(self, prefix: str | None, device: str = "cuda:0", n_threads: int = 0, _mem=None, sa_intv: int | None = 1):
        self.L = load_library()
        self.dev = torch.device(device)
        if _mem is not None:                                  # (index, contigs, packed reference) already in memory: from_memory()
            idx, self.contigs, self.pac = _mem
            self.l_pac = sum(c[1] for c in self.contigs)
            self.alt = np.ascontiguousarray([1 if (len(c) > 2 and c[2]) else 0 for c in self.contigs], dtype=np.uint8)
            self.contigs = [(c[0], c[1]) for c in self.contigs]
        else:
            idx = fmindex.read_index(prefix)
            self.contigs, self.l_pac = read_ann(prefix)
            self.alt = read_alt(prefix, self.contigs)
            pac = np.fromfile(prefix + ".pac", dtype=np.uint8)
            self.pac = np.ascontiguousarray(np.concatenate([pac[: (self.l_pac + 3) // 4], np.zeros(2, np.uint8)]))
        self.index = Index.upload(idx, pac=self.pac, l_pac=self.l_pac)
        if sa_intv:                                            # denser suffix-array samples than the files hold (every 16th row): ms on the device
            self.index.densify_sa(sa_intv)
        self.copt = ChainOpt(); self.L.bmh_chain_opt_default(C.byref(self.copt))
        self.ep = ExtParams.default()
        self.po = PostOpt(); self.L.bmh_post_opt_default(C.byref(self.po))
        self.pe = PeOpt(); self.L.bmh_pe_opt_default(C.byref(self.pe))
        # ALT contigs (<prefix>.alt): the chain filter, the marking of primary hits, MAPQ, the XA / pa tags and the clipping depend on them
        # (src/bwamem.c:446,518,571-574,702,714-760,1540,1663,1742,1755); the region tail of such an index runs on the host
        self.has_alt = bool(self.alt.any())
        if self.has_alt:
            self.copt.contig_is_alt = self.alt.ctypes.data; self.po.contig_is_alt = self.alt.ctypes.data
        self.n_threads = n_threads or int(self.L.bmh_effective_cpus())          # (the CPUs this process is granted, not the ones the machine shows)
        self.profile = bool(os.environ.get("BMH_ALIGNER_PROFILE"))
        self.c_off = np.ascontiguousarray(np.concatenate([[0], np.cumsum([c[1] for c in self.contigs])]), dtype=np.int64)

    @classmethod
    def from_memory(cls, idx, genome_fwd: np.ndarray, contigs=None, **kw) -> "Aligner":
        """idx: fmindex.FMDIndex of genome_fwd (nt4 codes); contigs: [(name, length)], default one sequence chrS"""
        pad = (-len(genome_fwd)) % 4
        codes = np.concatenate([genome_fwd, np.zeros(pad, np.uint8)]).reshape(-1, 4)
        pac = ((codes[:, 0] << 6) | (codes[:, 1] << 4) | (codes[:, 2] << 2) | codes[:, 3]).astype(np.uint8)
        pac = np.ascontiguousarray(np.concatenate([pac, np.zeros(2, np.uint8)]))
        return cls(None, _mem=(idx, contigs or [("chrS", int(len(genome_fwd)))], pac), **kw)

    def set_options(self, argv) -> None:
        """gase_aln's command-line options (src/fastmap.c:166-262) that reach this path, as a list of strings, e.g.
        ["-k", "23", "-A", "2", "-a"].  The device extension takes -A -B and the DELETION penalties -O -E for both gap
        kinds, as the reference's GPU extension does (src/fastmap.c:417-424); -O/-E given as "del,ins" keep the pair for
        the host stages.  Unlike the reference (update_a, src/fastmap.c), nothing is rescaled by -A: pass every value you want
        changed (-B -O -E -T -U).
        -d and -L are accepted and ignored (they do not reach the reference's GPU extension either).  Not modelled: -x -r -s
        -y (seeding variants the GPU seeding of the reference ignores too), -I -H -C -V."""
        import math
        co, ep, po, pe = self.copt, self.ep, self.po, self.pe
        i = 0
        def pair(v):
            a, _, b = v.replace(";", ",").partition(",")
            return int(a), int(b) if b else int(a)
        while i < len(argv):
            f = argv[i]
            if f in ("-a", "-M", "-Y", "-S", "-P", "-j"):
                if f == "-j":                             # the .alt file is ignored (src/fastmap.c:186,390-392): every sequence belongs to the primary assembly
                    self.alt[:] = 0; self.has_alt = False; co.contig_is_alt = None; po.contig_is_alt = None
                    if getattr(self, "_native", None) is not None: self._native.free(); self._native = None
                    c = getattr(self, "_cw_cache", None)         # a cached chain workspace still holds the ALT table (set_alt): the batch-by-batch path must not filter chains with it
                    if c is not None: c[0].free(); self._cw_cache = None
                elif f == "-a": po.flag_all = 1
                elif f == "-M": po.no_multi = 1
                elif f == "-Y": po.softclip = 1
                elif f == "-S": pe.no_rescue = 1
                else: pe.no_pairing = 1
                i += 1; continue
            if i + 1 >= len(argv):
                raise ValueError(f"option {f} needs a value")
            v = argv[i + 1]; i += 2
            if f == "-k": co.min_seed_len = int(v)
            elif f == "-w": co.w = int(v)
            elif f == "-c": co.max_occ = int(v)
            elif f == "-D": co.drop_ratio = float(v)
            elif f == "-G": co.max_chain_gap = int(v)
            elif f == "-N": co.max_chain_extend = int(v)
            elif f == "-W": co.min_chain_weight = int(v)
            elif f == "-X": co.mask_level = float(v)
            elif f == "-A": co.a = ep.a = int(v)
            elif f == "-B": co.b = ep.b = int(v)
            elif f == "-O": co.o_del, co.o_ins = pair(v); ep.o_del, ep.o_ins = co.o_del, co.o_ins
            elif f == "-E": co.e_del, co.e_ins = pair(v); ep.e_del, ep.e_ins = co.e_del, co.e_ins
            elif f == "-T": po.T = int(v)
            elif f == "-h": po.max_XA_hits, po.max_XA_hits_alt = pair(v)
            elif f == "-Q": po.mapQ_coef_len = float(int(v)); po.mapQ_coef_fac = int(math.log(int(v))) if int(v) > 0 else 0
            elif f == "-U": pe.pen_unpaired = int(v)
            elif f == "-m": pe.max_matesw = int(v)
            elif f == "-R":                                             # read group (bwa_set_rg, src/bwa.c:425-452): the header line and the records' RG:Z tag
                # bwa_escape (src/bwa.c:409-423): ONE pass from the left -- a backslash consumes the character behind it, which becomes a tab / newline /
                # carriage return / backslash for t / n / r / \\ and nothing at all otherwise ("\\\\t" is a backslash and a 't', not a backslash and a tab)
                if not v.startswith("@RG"):
                    raise ValueError("-R: the read group line must start with @RG")
                import re
                line = re.sub(r"\\(.?)", lambda m: {"t": "\t", "n": "\n", "r": "\r", "\\": "\\"}.get(m.group(1), ""), v, flags=re.S)
                if "\tID:" not in line:
                    raise ValueError("-R: the read group line must hold an ID")
                rid = re.split("[\t\n]", line.split("\tID:", 1)[1], 1)[0]            # the ID ends at the first tab or newline (src/bwa.c:440-446)
                if len(rid) > 255:
                    raise ValueError("-R: @RG:ID is longer than 255 characters")
                self.rg_line = line; self._rg_id = rid.encode(); po.rg_id = self._rg_id
                if getattr(self, "_native", None) is not None: self._native.free(); self._native = None
            elif f == "-t": self.ref_threads = int(v)                   # the reference cuts its batches at chunk_size * n_threads bases (align_file)
            elif f == "-K": self.ref_chunk_bases = int(v)               # ... or at this fixed size
            elif f in ("-l", "-v", "-f", "-d", "-L"): pass              # bookkeeping; -d -L: no effect on the GPU extension
            else:
                raise ValueError(f"option {f} is not modelled")

    def header(self) -> str:
        return "".join(f"@SQ\tSN:{n}\tLN:{l}\n" for n, l in self.contigs) + (getattr(self, "rg_line", "") + "\n" if getattr(self, "rg_line", "") else "")

    def align_batch(self, names, seqs=None, id0: int = 0, paired: bool = False, as_bytes: bool = False):
        """SAM records of one batch of reads: a ReadSet, or (names, seqs) lists of str / ASCII uint8 arrays; id0 = index of
        its first read in the run.  paired: interleaved pairs (gase_aln -p); the insert-size statistics are the batch's."""
        rs = names if isinstance(names, ReadSet) else ReadSet.from_lists(names, seqs)
        L, dev, n = self.L, self.dev, len(rs)
        self._as_bytes = as_bytes
        if n == 0:
            return (np.zeros(0, np.uint8) if as_bytes == "view" else b"") if as_bytes else ""
        names = (rs.name_blob, rs.name_off)
        import time
        _t = [time.perf_counter()]; _nm = []
        def _lap(name):
            if self.profile:
                torch.cuda.synchronize(); _t.append(time.perf_counter()); _nm.append(name)
        lens, offs, ascii_ = rs.lens, np.ascontiguousarray(rs.offs), rs.ascii
        # the device job builder takes reads of up to 700 bases (CH_MAX_READ_LEN, csrc/chain_core.h: the extension kernels' classes end at
        # 768 columns), the reference's seed filter mem_flt_chained_seeds (src/bwamem.c:970-991) included when a small -W makes it apply;
        # a batch with a longer read (where the filter applies without -W, beyond ~730 bp) goes through bmh_build_jobs
        host_jobs = bool((lens > 700).any())
        if int(lens.sum()) >= 1 << 31:
            raise ValueError("a batch holds 2^31 bases or more: offsets inside a batch are 32-bit (use a smaller batch_reads)")
        codes = rs.codes if getattr(rs, "codes", None) is not None else _NT4[ascii_]
        a_c = np.ascontiguousarray(ascii_)
        r = torch.from_numpy(a_c if a_c.flags.writeable else a_c.copy()).to(dev)        # (no copy of 150 MB per million reads unless needed)
        o = torch.from_numpy(offs.astype(np.int64)).to(torch.int32).to(dev)
        l = torch.from_numpy(lens.astype(np.int64)).to(torch.int32).to(dev)
        _lap("host prep + H2D")
        nb = max(int(lens.sum()), 1)
        ws = self._seed_ws(n, nb)
        _lap("seed workspace")
        s = ws.seed_batch(self.index, r, o, l, self.copt.min_seed_len)
        _lap("seeding")
        e = self.ep                                             # the reference's GPU extension: deletion penalties for both gap kinds
        ext_p = ExtParams(e.a, e.b, e.o_del, e.e_del, e.o_del, e.e_del, e.zdrop, e.end_bonus)
        cw = None
        if host_jobs:
            from .lib import HostJobs, seeds_to_host, extend_batch
            hj = HostJobs(self.l_pac, codes, offs, lens, seeds_to_host(s, n), n_threads=self.n_threads, opt=self.copt,
                          contigs=self.contigs if len(self.contigs) > 1 else None, pac=self.pac)
            nr, nj = hj.n_regs, hj.n_jobs
            _lap("chain (host builder: a read beyond 700 bp)")
            out3 = torch.zeros(max(nj, 1), 3, dtype=torch.int32, device=dev)
            if nj:
                d = [torch.from_numpy(np.ascontiguousarray(x).view(np.int32) if x.dtype == np.uint32 else np.ascontiguousarray(x)).to(dev) for x in hj.jobs()]
                extend_batch(*d, out3, params=ext_p)
                # a flank longer than the DP kernels take (768 query bases: a read beyond ~790 bp seeded near one end) comes back as
                # INT32_MIN and must never reach the merge; the device builder refuses such reads itself
                n_bad = int(L.bmh_extend_last_unsupported())
                if n_bad:
                    hj.free()
                    raise NotImplementedError(f"{n_bad} extension job(s) of this batch have a query side longer than 768 bases: reads this long are beyond the "
                                              "extension kernels (the reference's own GASAL2 build is sized by MAX_SEQ_LEN, README.md:38)")
            regs_h = np.ascontiguousarray(hj.merge(out3[:nj].cpu().numpy())) if nr else np.zeros((0, 8), np.int32)
            rpr_h = np.ascontiguousarray(hj.regs_per_read.copy()); fr_h = np.ascontiguousarray(hj.frac_rep(), dtype=np.float32)
            hj.free()
            _lap("extend+merge")
        else:
            cw = self._chain_ws(n, max(int(s.n_seeds), 1))
            dj = cw.chain_batch(self.index, r, o, l, s)
            nr, nj = int(dj.n_regs), int(dj.n_jobs)
            out3 = torch.zeros(max(nj, 1), 3, dtype=torch.int32, device=dev)
            regs = torch.zeros(max(nr, 1), 8, dtype=torch.int32, device=dev)
            _lap("chain")
            cw.extend(out3, params=ext_p)
            cw.merge(out3, regs)
            _lap("extend+merge")
            if not paired and not self.has_alt:
                # single-end: the region tail runs on the device too (bmh_finalize_regs_device); what comes back over PCIe are the records
                po = PostOpt.from_buffer_copy(self.po); po.id0 = id0
                from .lib import CapacityError, finalize_regs_device
                try:
                    d_fin, d_opr = finalize_regs_device(self.index, self.copt, self.ep, po, r, o, regs, nr, dj.d_regs_per_read, dj.d_frac_rep, n,
                                                        contigs=self.contigs if len(self.contigs) > 1 else None)
                except CapacityError:
                    # a read beyond the device tail's fixed limits (65 535 near-equal regions, a patch alignment of more than 1 022
                    # bases): this batch takes the host form below, which has none of them -- same records
                    d_fin = None
                    self.host_tail_batches = getattr(self, "host_tail_batches", 0) + 1
                if d_fin is not None:
                    _lap("finalize (device)")
                    fin = self._d2h("fin", d_fin); opr = np.ascontiguousarray(d_opr.cpu().numpy().view(np.uint32)[:n]); m = len(fin)
                    _lap("D2H records")
                    self._lap = _lap; self._prof = (_t, _nm)
                    return self._finish_single(names, codes, offs, lens, r, o, l, fin, opr, m, po, cw, ws, _lap, _t, _nm, as_bytes, fin_t=d_fin)
            rpr = torch.empty(n, dtype=torch.int32, device=dev); fr = torch.empty(n, dtype=torch.float32, device=dev)
            _memcpy_d2d(rpr.data_ptr(), dj.d_regs_per_read, 4 * n); _memcpy_d2d(fr.data_ptr(), dj.d_frac_rep, 4 * n)
            regs_h = np.ascontiguousarray(regs[:nr].cpu().numpy())
            rpr_h = np.ascontiguousarray(rpr.cpu().numpy().view(np.uint32)); fr_h = np.ascontiguousarray(fr.cpu().numpy())
            _lap("D2H regions")
        po = PostOpt.from_buffer_copy(self.po); po.id0 = id0
        self._lap = _lap; self._prof = (_t, _nm)
        if paired:
            return self._finish_pairs(names, codes, offs, lens, r, o, l, regs_h, rpr_h, fr_h, po, cw, ws)
        fin = np.zeros((max(nr, 1), 16), np.int32); opr = np.zeros(n, np.uint32)
        m = L.bmh_finalize_regs(C.byref(self.copt), C.byref(self.ep), C.byref(po), self.l_pac, _np_ptr(self.pac, _u8p), n, _np_ptr(codes, _u8p),
                                _np_ptr(offs, _u64p), _np_ptr(regs_h, _i32p), _np_ptr(rpr_h, _u32p), fr_h.ctypes.data_as(C.POINTER(C.c_float)),
                                len(self.contigs), self.c_off.ctypes.data_as(C.c_void_p), _np_ptr(fin, _i32p), _np_ptr(opr, _u32p), self.n_threads)
        if m < 0:
            raise RuntimeError("bmh_finalize_regs: " + (L.bmh_last_error() or b"").decode())
        fin = np.ascontiguousarray(fin[:m])
        _lap("finalize (host)")
        return self._finish_single(names, codes, offs, lens, r, o, l, fin, opr, m, po, cw, ws, _lap, _t, _nm, as_bytes)

    def _finish_single(self, names, codes, offs, lens, r, o, l, fin, opr, m, po, cw, ws, _lap, _t, _nm, as_bytes, fin_t=None):
        """records of the region tail -> CIGARs of the ones the formatter needs (device) -> SAM text (host)"""
        L, n = self.L, len(lens)
        need = np.zeros(max(m, 1), np.uint8)
        L.bmh_sam_need_cigar(C.byref(po), _np_ptr(fin, _i32p), _np_ptr(opr, _u32p), n, _np_ptr(need, _u8p))
        sel = np.nonzero(need[:m])[0].astype(np.int32)
        slot = np.full(max(m, 1), -1, np.int64); slot[sel] = np.arange(len(sel))
        aln_h, cg_h, md_h = self._cigars(r, o, l, fin, sel, fin_t=fin_t)
        _lap("cigar + D2H")
        txt = format_sam(po, names, codes, offs, lens, self.contigs, fin if m else np.zeros((1, 16), np.int32), opr, slot, aln_h, cg_h, md_h,
                         as_bytes=as_bytes)
        _lap("format (host)")
        if self.profile:
            import sys
            sys.stderr.write("[aligner] " + ", ".join("%s %.1f ms" % (nm, (_t[i + 1] - _t[i]) * 1e3) for i, nm in enumerate(_nm)) + "\n")
        return txt

    def _cigars(self, r, o, l, fin, sel, fin_t=None):
        """bmh_cigar_batch for the selected records with compact buffers; the few records that overflow them (flag 1: more
        ops, flag 8: longer MD) are redone with large ones and patched in"""
        dev = self.dev
        max_cigar, md_cap = 64, 1024
        if not len(sel):
            return np.zeros((1, 8), np.int32), np.zeros((1, max_cigar), np.uint32), np.zeros((1, md_cap), np.uint8)
        if fin_t is None or not len(fin_t):
            fin_t = torch.from_numpy(fin.copy()).to(dev)
        cg, aln, md = cigar_batch(self.index, r, o, l, fin_t, len(sel), sel_t=torch.from_numpy(sel).to(dev), params=self.ep, opt_w=self.copt.w,
                                  max_cigar=16, md_cap=96)
        aln_h = self._d2h("aln", aln)
        cg_h = self._d2h("cg", cg).view(np.uint32); md_h = self._d2h("md", md)
        over = np.flatnonzero(aln_h[:, 7] & 9)
        if over.size:
            cg_c, md_c = cg_h, md_h
            cg_h = np.zeros((len(sel), max_cigar), np.uint32); md_h = np.zeros((len(sel), md_cap), np.uint8)
            cg_h[:, :16] = cg_c; md_h[:, :96] = md_c
            cg2, aln2, md2 = cigar_batch(self.index, r, o, l, fin_t, len(over), sel_t=torch.from_numpy(sel[over].copy()).to(dev), params=self.ep,
                                         opt_w=self.copt.w, max_cigar=max_cigar, md_cap=md_cap)
            aln_h[over] = aln2.cpu().numpy(); cg_h[over] = cg2.cpu().numpy().view(np.uint32); md_h[over] = md2.cpu().numpy()
        if (aln_h[:, 7] & ~2).any():
            raise RuntimeError("bmh_cigar_batch flagged an alignment (CIGAR or MD longer than the buffers)")
        return aln_h, cg_h, md_h

    def _finish_pairs(self, names, codes, offs, lens, r, o, l, regs_h, rpr_h, fr_h, po, cw, ws) -> str:
        L, dev, n = self.L, self.dev, len(lens)
        # (the mate rescue's local alignments go to the device as one batch: bmh_finalize_pairs_dev)
        fin, opr, h_rec, unflag, _ = finalize_pairs(self.copt, self.ep, po, self.l_pac, self.pac, codes, offs, lens, regs_h, rpr_h, fr_h,
                                                    contigs=self.contigs if len(self.contigs) > 1 else None, n_threads=self.n_threads, pe=self.pe,
                                                    device=(self.index, r, o, None))
        fin = np.ascontiguousarray(fin); m = len(fin)
        need = np.zeros(max(m, 1), np.uint8)
        L.bmh_sam_need_cigar_pe(C.byref(po), _np_ptr(fin if m else np.zeros((1, 16), np.int32), _i32p), _np_ptr(np.ascontiguousarray(opr), _u32p),
                                _np_ptr(np.ascontiguousarray(h_rec), _i32p), n, _np_ptr(need, _u8p))
        sel = np.nonzero(need[:m])[0].astype(np.int32)
        slot = np.full(max(m, 1), -1, np.int64); slot[sel] = np.arange(len(sel))
        aln_h, cg_h, md_h = self._cigars(r, o, l, fin, sel)
        txt = format_sam(po, names, codes, offs, lens, self.contigs, fin if m else np.zeros((1, 16), np.int32), opr, slot, aln_h, cg_h, md_h,
                         h_rec=h_rec, unflag=unflag, as_bytes=self._as_bytes)
        return txt

    def _native_aligner(self):
        """the bmh_aligner_t of this aligner's index and options (made again when set_options changed them)"""
        from .lib import NativeAligner
        nat = getattr(self, "_native", None)
        opts = (bytes(self.copt), bytes(self.ep), bytes(self.po), bytes(self.pe))
        if nat is not None and nat.options != opts:
            nat.free(); nat = None
        if nat is None:
            nat = self._native = NativeAligner(self.index, self.pac, self.l_pac, self.contigs, self.alt if self.has_alt else None, self.copt, self.ep, self.po, self.pe)
        return nat

    def align_file(self, reads_fa: str, out, batch_reads: int = 0, paired: bool = False, chunk_bases: int = 0) -> int:
        """out: a text or binary file object.  Batches are cut the way the reference's bseq_read cuts them (src/bwa.c, called with
        chunk_size * n_threads = 10 Mbases per thread, or -K, src/fastmap.c:527): reads are added until the batch holds at least
        chunk_bases bases and an even number of reads -- in paired mode the insert-size statistics are those of the batch, so the
        cut decides flags and MAPQ of borderline pairs.  chunk_bases 0: 10 000 000 x the thread count given with -t (1).
        batch_reads > 0 cuts by read count instead (the earlier behaviour)."""
        binary = "b" in getattr(out, "mode", "") or hasattr(out, "getbuffer")
        # The file batch by batch through bmh_aligner_run_fasta (a loader thread cuts and fills batch k+1 .. while the lanes are on batch k: nothing of the file is
        # held beyond the batches in flight), when every read fits the device job builder (bmh_fasta_scan: one counting pass); BMH_ALIGNER_STREAM=0: load the
        # whole file first (below).
        if os.environ.get("BMH_ALIGNER_NATIVE", "1") != "0" and os.environ.get("BMH_ALIGNER_STREAM", "1") != "0" and not self.profile:
            from .lib import NativeAligner, fasta_scan
            info = fasta_scan(reads_fa)
            if info["n_reads"] and info["max_len"] <= 700:
                out.write(self.header().encode() if binary else self.header())
                cb = 0
                if batch_reads <= 0:
                    cb = chunk_bases or int(getattr(self, "ref_chunk_bases", 0)) or 10_000_000 * max(1, int(getattr(self, "ref_threads", 1)))
                    if not paired and not chunk_bases:
                        cb = max(cb, 150_000_000)             # (single-end records do not depend on the cuts: see below)
                    cb = min(cb, (1 << 31) - 1024)
                nat = self._native_aligner()
                self.last_stats = nat.run_fasta(reads_fa, paired, (lambda mv: out.write(mv)) if binary else (lambda mv: out.write(bytes(mv).decode())),
                                                batch_bases=cb, batch_reads=max(batch_reads, 0),
                                                n_lanes=int(os.environ.get("BMH_ALIGNER_LANES", "3" if paired else "2")), n_threads=self.n_threads)
                return info["n_reads"]
        rs = read_fasta_reads(reads_fa)
        out.write(self.header().encode() if binary else self.header())
        n = len(rs)
        if batch_reads > 0:
            if paired:
                batch_reads -= batch_reads & 1
            cuts = list(range(0, n, batch_reads)) + [n]
        else:
            cb = chunk_bases or int(getattr(self, "ref_chunk_bases", 0)) or 10_000_000 * max(1, int(getattr(self, "ref_threads", 1)))
            if not paired and not chunk_bases:
                # single-end records do not depend on where the batches are cut (the tie-break hash takes the read's index in the run), and a
                # batch of a quarter of a million reads spends a third of its time in fixed per-batch latencies: at least 150 Mbases a batch
                cb = max(cb, 150_000_000)
            cb = min(cb, (1 << 31) - 1024)                               # offsets inside a batch are 32-bit
            csum = np.cumsum(rs.lens.astype(np.int64))
            cuts, b = [0], 0
            while b < n:
                base = csum[b - 1] if b else 0
                e = int(np.searchsorted(csum, base + cb, side="left")) + 1      # first read count whose bases reach chunk_bases
                e += (e - b) & 1                                             # ... and is even (bseq_read: size >= chunk_size && (n & 1) == 0)
                e = min(e, n)
                cuts.append(e); b = e
        # The native pipeline (csrc/align_pipeline.hip: batches driven by C threads, the text of batch k formatted while batches
        # k+1, k+2 are on the device) takes every read set whose reads fit the device job builder; BMH_ALIGNER_NATIVE=0 keeps the
        # batch-after-batch Python loop below (the same stages through the same entry points: the cross-check of the native one).
        if n and os.environ.get("BMH_ALIGNER_NATIVE", "1") != "0" and rs.codes is not None and int(rs.lens.max()) <= 700 and not self.profile:
            nat = self._native_aligner()
            self.last_stats = nat.run(rs, cuts, paired, (lambda mv: out.write(mv)) if binary else (lambda mv: out.write(bytes(mv).decode())),
                                      n_lanes=int(os.environ.get("BMH_ALIGNER_LANES", "3" if paired else "2")), n_threads=self.n_threads)   # (pairs: a lane waits for host walks in the middle of its batch)
            return n
        for b, e in zip(cuts[:-1], cuts[1:]):
            if e > b:
                out.write(self.align_batch(rs.slice(b, e), id0=b, paired=paired, as_bytes="view" if binary else False))   # (binary: the library's buffer, uncopied)
        return n

    # The seeding and chaining workspaces are kept between batches (allocating and freeing a few GB of HBM per batch cost more than
    # the kernels of an easy batch); a batch that needs more gets a new one, a quarter larger than it asked for.
    def _d2h(self, key: str, t):
        """device tensor -> numpy array in a PINNED host buffer kept per key (one DMA instead of a staged pageable copy: the CIGAR /
        MD / record arrays of a million reads are 200 MB).  The array is valid until the next call with the same key."""
        t = t.contiguous()
        nbytes = t.numel() * t.element_size()
        if nbytes == 0:
            return t.cpu().numpy()
        c = getattr(self, "_pin", None)
        if c is None:
            c = self._pin = {}
        buf = c.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = c[key] = torch.empty(nbytes + nbytes // 4, dtype=torch.uint8, pin_memory=True)
        dst = buf[:nbytes].view(t.dtype).view(t.shape)
        dst.copy_(t, non_blocking=True)
        torch.cuda.current_stream(t.device).synchronize()
        return dst.numpy()

    def _seed_ws(self, n: int, nb: int):
        c = getattr(self, "_ws_cache", None)
        if c is not None and n <= c[1] and nb <= c[2]:
            return c[0]
        if c is not None:
            c[0].free()
        cn, cb = (n, nb) if c is None else (n + n // 4, nb + nb // 4)
        ws = SeedWorkspace(cn, cb, max_cands=cb, max_occ=max(64 * cn, 1 << 16))       # one candidate per base is the hard upper bound
        self._ws_cache = (ws, cn, cb)
        return ws

    def _chain_ws(self, n: int, n_seeds: int):
        c = getattr(self, "_cw_cache", None)
        if c is not None and n <= c[1] and n_seeds <= c[2]:
            return c[0]
        if c is not None:
            c[0].free()
        cn, cs = (n, n_seeds + n_seeds // 8) if c is None else (n + n // 4, n_seeds + n_seeds // 4)
        cw = ChainWorkspace(cn, cs, opt=self.copt)
        cw.set_materialize(False)
        if len(self.contigs) > 1:
            cw.set_contigs(self.contigs)
            if self.has_alt:
                cw.set_alt(self.alt)
        self._cw_cache = (cw, cn, cs)
        return cw

    def __del__(self):                      # an Aligner dropped without close() must not keep its workspaces in HBM
        try:
            for nm in ("_ws_cache", "_cw_cache"):
                c = getattr(self, nm, None)
                if c is not None:
                    c[0].free(); setattr(self, nm, None)
        except Exception:
            pass

    def close(self):
        nat = getattr(self, "_native", None)
        if nat is not None:
            nat.free(); self._native = None
        for nm in ("_ws_cache", "_cw_cache"):
            c = getattr(self, nm, None)
            if c is not None:
                c[0].free(); setattr(self, nm, None)
        self.index.free()
