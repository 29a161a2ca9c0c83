"""Single-end `gase_aln` on the device-resident path: index files + FASTA reads -> SAM.

The reference's CLI keeps working on this library through the drop-in headers (INTEGRATION.md sections 1-2); this module
is the same job done through the device-level C ABI instead -- seeding, chaining / job construction, extension and the
region merge on the GPU (bmh_seed_batch, bmh_chain_batch, bmh_chain_extend, bmh_chain_merge), the region tail on host
threads as in the reference (bmh_finalize_regs), CIGAR / NM / MD on the GPU (bmh_cigar_batch), SAM text on the host
(bmh_format_sam).  It writes the records the reference writes, byte for byte (tests/test_gpu_parity.py).  torch is used
for device memory only.  Interleaved pairs (gase_aln -p) go through bmh_finalize_pairs / bmh_format_sam_pe; their
insert-size statistics are per batch as in the reference, so identical output needs the reference's batching (one batch
here = batch_reads reads).  ALT contigs, read groups and FASTQ qualities are not handled.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import fmindex
from .lib import (ChainOpt, ChainWorkspace, ExtParams, Index, PostOpt, SeedWorkspace, _memcpy_d2d, _np_ptr, _i32p, _u32p, _u64p, _u8p,
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


def read_fasta_reads(path: str):
    names, seqs = [], []
    with open(path, "rb") as f:
        for line in f:
            line = line.rstrip(b"\r\n")
            if not line:
                continue
            if line[:1] == b">":
                names.append(line[1:].split()[0].decode())
            else:
                seqs.append(np.frombuffer(line, dtype=np.uint8))
    return names, seqs


class Aligner:
    def __init__(self, prefix: str, device: str = "cuda:0", n_threads: int = 0):
        self.L = load_library()
        self.dev = torch.device(device)
        idx = fmindex.read_index(prefix)
        self.contigs, self.l_pac = read_ann(prefix)
        pac = np.fromfile(prefix + ".pac", dtype=np.uint8)
        self.pac = np.ascontiguousarray(np.concatenate([pac[: (self.l_pac + 3) // 4], np.zeros(2, np.uint8)]))
        self.index = Index.upload(idx, pac=self.pac, l_pac=self.l_pac)
        self.copt = ChainOpt(); self.L.bmh_chain_opt_default(C.byref(self.copt))
        self.ep = ExtParams.default()
        self.n_threads = n_threads or (os.cpu_count() or 1)
        self.c_off = np.ascontiguousarray(np.concatenate([[0], np.cumsum([c[1] for c in self.contigs])]), dtype=np.int64)

    def header(self) -> str:
        return "".join(f"@SQ\tSN:{n}\tLN:{l}\n" for n, l in self.contigs)

    def align_batch(self, names, seqs, id0: int = 0, paired: bool = False) -> str:
        """SAM records of one batch of reads (ASCII uint8 arrays); id0 = index of its first read in the run.
        paired: the batch holds interleaved pairs (gase_aln -p); the insert-size statistics are those of the batch."""
        L, dev, n = self.L, self.dev, len(seqs)
        if n == 0:
            return ""
        lens = np.array([len(s) for s in seqs], np.uint32)
        offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
        ascii_ = np.concatenate(seqs) if lens.sum() else np.zeros(1, np.uint8)
        codes = _NT4[ascii_]
        r = torch.from_numpy(ascii_.copy()).to(dev)
        o = torch.from_numpy(offs.astype(np.int64)).to(torch.int32).to(dev)
        l = torch.from_numpy(lens.astype(np.int64)).to(torch.int32).to(dev)
        ws = SeedWorkspace(n, max(int(lens.sum()), 1))
        s = ws.seed_batch(self.index, r, o, l, self.copt.min_seed_len)
        cw = ChainWorkspace(n, max(int(s.n_seeds), 1), opt=self.copt)
        cw.set_materialize(False)
        if len(self.contigs) > 1:
            cw.set_contigs(self.contigs)
        dj = cw.chain_batch(self.index, r, o, l, s)
        nr, nj = int(dj.n_regs), int(dj.n_jobs)
        out3 = torch.zeros(max(nj, 1), 3, dtype=torch.int32, device=dev)
        regs = torch.zeros(max(nr, 1), 8, dtype=torch.int32, device=dev)
        cw.extend(out3, params=self.ep)
        cw.merge(out3, regs)
        rpr = torch.empty(n, dtype=torch.int32, device=dev); fr = torch.empty(n, dtype=torch.float32, device=dev)
        _memcpy_d2d(rpr.data_ptr(), dj.d_regs_per_read, 4 * n); _memcpy_d2d(fr.data_ptr(), dj.d_frac_rep, 4 * n)
        regs_h = np.ascontiguousarray(regs[:nr].cpu().numpy())
        rpr_h = np.ascontiguousarray(rpr.cpu().numpy().view(np.uint32)); fr_h = np.ascontiguousarray(fr.cpu().numpy())
        po = PostOpt(); L.bmh_post_opt_default(C.byref(po)); po.id0 = id0
        if paired:
            return self._finish_pairs(names, codes, offs, lens, r, o, l, regs_h, rpr_h, fr_h, po, cw, ws)
        fin = np.zeros((max(nr, 1), 16), np.int32); opr = np.zeros(n, np.uint32)
        m = L.bmh_finalize_regs(C.byref(self.copt), C.byref(self.ep), C.byref(po), self.l_pac, _np_ptr(self.pac, _u8p), n, _np_ptr(codes, _u8p),
                                _np_ptr(offs, _u64p), _np_ptr(regs_h, _i32p), _np_ptr(rpr_h, _u32p), fr_h.ctypes.data_as(C.POINTER(C.c_float)),
                                len(self.contigs), self.c_off.ctypes.data_as(C.c_void_p), _np_ptr(fin, _i32p), _np_ptr(opr, _u32p), self.n_threads)
        if m < 0:
            raise RuntimeError("bmh_finalize_regs: " + (L.bmh_last_error() or b"").decode())
        fin = np.ascontiguousarray(fin[:m])
        need = np.zeros(max(m, 1), np.uint8)
        L.bmh_sam_need_cigar(C.byref(po), _np_ptr(fin, _i32p), _np_ptr(opr, _u32p), n, _np_ptr(need, _u8p))
        sel = np.nonzero(need[:m])[0].astype(np.int32)
        slot = np.full(max(m, 1), -1, np.int64); slot[sel] = np.arange(len(sel))
        max_cigar, md_cap = 64, 1024
        if len(sel):
            cg, aln, md = cigar_batch(self.index, r, o, l, torch.from_numpy(fin.copy()).to(dev), len(sel), sel_t=torch.from_numpy(sel).to(dev),
                                      params=self.ep, opt_w=self.copt.w, max_cigar=max_cigar, md_cap=md_cap)
            aln_h = aln.cpu().numpy(); cg_h = cg.cpu().numpy().view(np.uint32); md_h = md.cpu().numpy()
            if (aln_h[:, 7] & ~2).any():
                raise RuntimeError("bmh_cigar_batch flagged an alignment (CIGAR or MD longer than the buffers)")
        else:
            aln_h = np.zeros((1, 8), np.int32); cg_h = np.zeros((1, max_cigar), np.uint32); md_h = np.zeros((1, md_cap), np.uint8)
        txt = format_sam(po, names, codes, offs, lens, self.contigs, fin if m else np.zeros((1, 16), np.int32), opr, slot, aln_h, cg_h, md_h)
        cw.free(); ws.free()
        return txt

    def _finish_pairs(self, names, codes, offs, lens, r, o, l, regs_h, rpr_h, fr_h, po, cw, ws) -> str:
        L, dev, n = self.L, self.dev, len(lens)
        fin, opr, h_rec, unflag, _ = finalize_pairs(self.copt, self.ep, po, self.l_pac, self.pac, codes, offs, lens, regs_h, rpr_h, fr_h,
                                                    contigs=self.contigs if len(self.contigs) > 1 else None, n_threads=self.n_threads)
        fin = np.ascontiguousarray(fin); m = len(fin)
        need = np.zeros(max(m, 1), np.uint8)
        L.bmh_sam_need_cigar_pe(C.byref(po), _np_ptr(fin if m else np.zeros((1, 16), np.int32), _i32p), _np_ptr(np.ascontiguousarray(opr), _u32p),
                                _np_ptr(np.ascontiguousarray(h_rec), _i32p), n, _np_ptr(need, _u8p))
        sel = np.nonzero(need[:m])[0].astype(np.int32)
        slot = np.full(max(m, 1), -1, np.int64); slot[sel] = np.arange(len(sel))
        max_cigar, md_cap = 64, 1024
        if len(sel):
            cg, aln, md = cigar_batch(self.index, r, o, l, torch.from_numpy(fin.copy()).to(dev), len(sel), sel_t=torch.from_numpy(sel).to(dev),
                                      params=self.ep, opt_w=self.copt.w, max_cigar=max_cigar, md_cap=md_cap)
            aln_h = aln.cpu().numpy(); cg_h = cg.cpu().numpy().view(np.uint32); md_h = md.cpu().numpy()
            if (aln_h[:, 7] & ~2).any():
                raise RuntimeError("bmh_cigar_batch flagged an alignment (CIGAR or MD longer than the buffers)")
        else:
            aln_h = np.zeros((1, 8), np.int32); cg_h = np.zeros((1, max_cigar), np.uint32); md_h = np.zeros((1, md_cap), np.uint8)
        txt = format_sam(po, names, codes, offs, lens, self.contigs, fin if m else np.zeros((1, 16), np.int32), opr, slot, aln_h, cg_h, md_h,
                         h_rec=h_rec, unflag=unflag)
        cw.free(); ws.free()
        return txt

    def align_file(self, reads_fa: str, out, batch_reads: int = 500_000, paired: bool = False) -> int:
        names, seqs = read_fasta_reads(reads_fa)
        out.write(self.header())
        if paired:
            batch_reads -= batch_reads & 1
        for b in range(0, len(seqs), batch_reads):
            out.write(self.align_batch(names[b:b + batch_reads], seqs[b:b + batch_reads], id0=b, paired=paired))
        return len(seqs)

    def close(self):
        self.index.free()
