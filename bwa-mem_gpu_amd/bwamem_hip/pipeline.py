"""Host-side pipeline pieces shared by bench.py, the multi-GPU launcher and the tests.

Nothing here computes the hot path: seeding and extension go through the C ABI
(lib.py -> libbwamem_hip.so).  torch is used for device memory, for the
bench-scale index build (fmindex) and for the vectorised construction of
extension jobs from seeds (the host "job builder" role of the reference,
src/bwamem.c:1170-1479, restated here only for the FIRST seed of a read --
see DESIGN.md "extension jobs in the bench").
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch

from . import fmindex, synth
from .lib import ExtParams, Index, SeedWorkspace, extend_batch


@dataclass
class DeviceReads:
    ascii: torch.Tensor   # uint8 [n_bases]
    offs: torch.Tensor    # int32 [n_reads]
    lens: torch.Tensor    # int32 [n_reads]
    codes: torch.Tensor   # uint8 [n_reads, L] nt4 codes (fixed length batches only)

    @property
    def n(self) -> int:
        return int(self.lens.numel())


def reads_to_device(reads2d: np.ndarray, device) -> DeviceReads:
    n, L = reads2d.shape
    codes = torch.from_numpy(np.ascontiguousarray(reads2d)).to(device)
    lut = torch.tensor(list(b"ACGTN"), dtype=torch.uint8, device=device)
    ascii_ = lut[codes.long()].reshape(-1).contiguous()
    offs = (torch.arange(n, device=device, dtype=torch.int64) * L).to(torch.int32)
    lens = torch.full((n,), L, dtype=torch.int32, device=device)
    return DeviceReads(ascii_, offs, lens, codes)


def index_to_device_tensors(idx: fmindex.FMDIndex, device):
    """Pack the index arrays into HBM tensors, blocks padded to whole 32-byte units."""
    nblk = (idx.seq_len + 63) // 64 + 1
    bwt = torch.zeros(nblk * 8, dtype=torch.int32, device=device)
    w = torch.from_numpy(idx.bwt_words.view(np.int32))
    bwt[: w.numel()] = w.to(device)
    sa = torch.from_numpy(idx.sa.view(np.int32)).to(device)
    bits = torch.from_numpy(idx.sa_bits.view(np.int32)).to(device)
    return bwt, sa, bits


def cal_max_gap(qlen: torch.Tensor, a=1, o_del=6, e_del=1, o_ins=6, e_ins=1, w=300) -> torch.Tensor:
    """src/bwamem.c:996-1002"""
    l_del = ((qlen * a - o_del).double() / e_del + 1.0).to(torch.int64)
    l_ins = ((qlen * a - o_ins).double() / e_ins + 1.0).to(torch.int64)
    l = torch.maximum(l_del, l_ins).clamp(min=1)
    return torch.minimum(l, torch.full_like(l, w << 1))


@dataclass
class ExtJobs:
    q: torch.Tensor; qoff: torch.Tensor; qlen: torch.Tensor
    t: torch.Tensor; toff: torch.Tensor; tlen: torch.Tensor
    h0: torch.Tensor
    read: torch.Tensor    # read index of each job
    side: torch.Tensor    # 0 = left, 1 = right

    @property
    def n(self) -> int:
        return int(self.qlen.numel())


def first_seed_jobs(seeds: dict, reads: DeviceReads, genome_fwd: torch.Tensor, w: int = 300) -> ExtJobs:
    """LEFT/RIGHT extension jobs of the longest seed (first occurrence) of every read, built the
    way mem_chain2aln builds them for the first seed of a chain (src/bwamem.c:1180-1201 rmax,
    :1315-1334 reversed left side, h0 = seed length :1360,1404).  seeds: torch tensors rbeg[int64],
    qbeg[int32 n,2], score[int32], n_ref_pos, prefix (device)."""
    dev = genome_fwd.device
    l_pac = int(genome_fwd.numel())
    L = reads.codes.shape[1]
    rbeg, qbeg, score = seeds["rbeg"], seeds["qbeg"].long(), seeds["score"].long()
    n_ref, prefix = seeds["n_ref_pos"].long(), seeds["prefix"].long()
    ns = int(rbeg.numel())
    if ns == 0:
        z = torch.zeros(0, dtype=torch.int32, device=dev)
        return ExtJobs(torch.zeros(1, dtype=torch.uint8, device=dev), z, z, torch.zeros(1, dtype=torch.uint8, device=dev), z, z, z, z, z)
    # group heads carry score > 0; longest seed per read = max (qe-qb), first on ties
    read_of = torch.repeat_interleave(torch.arange(reads.n, device=dev), n_ref)
    slen = qbeg[:, 1] - qbeg[:, 0]
    head = score > 0
    key = torch.where(head, slen * (ns + 1) + (ns - torch.arange(ns, device=dev)), torch.zeros_like(slen))
    best = torch.zeros(reads.n, dtype=torch.int64, device=dev).scatter_reduce(0, read_of, key, reduce="amax", include_self=True)
    has = best > 0
    pick = (ns - (best % (ns + 1)))[has]
    rd = torch.nonzero(has)[:, 0]
    rb, qb, ln = rbeg[pick], qbeg[pick, 0], slen[pick]
    qe = qb + ln
    # seeds bridging the forward/reverse boundary are discarded by the host (bns_intv2rid < 0)
    ok = ~((rb < l_pac) & (rb + ln > l_pac))
    rd, rb, qb, ln, qe = rd[ok], rb[ok], qb[ok], ln[ok], qe[ok]
    rem = L - qe
    r0 = rb - (qb + cal_max_gap(qb, w=w))
    r1 = rb + ln + (rem + cal_max_gap(rem, w=w))
    r0 = r0.clamp(min=0); r1 = r1.clamp(max=2 * l_pac)
    cross = (r0 < l_pac) & (l_pac < r1)
    r1 = torch.where(cross & (rb < l_pac), torch.full_like(r1, l_pac), r1)
    r0 = torch.where(cross & (rb >= l_pac), torch.full_like(r0, l_pac), r0)
    left = qb > 0
    right = rem > 0
    # job order: per read LEFT then RIGHT (fill order of src/bwamem.c:1352-1426)
    n_l, n_r = int(left.sum()), int(right.sum())
    qlen = torch.cat([qb[left], rem[right]])
    tlen = torch.cat([(rb - r0)[left], (r1 - (rb + ln))[right]])
    h0 = torch.cat([ln[left], ln[right]])
    jr = torch.cat([rd[left], rd[right]])
    side = torch.cat([torch.zeros(n_l, dtype=torch.int64, device=dev), torch.ones(n_r, dtype=torch.int64, device=dev)])
    order = torch.argsort(jr * 2 + side)
    qlen, tlen, h0, jr, side = qlen[order], tlen[order], h0[order], jr[order], side[order]
    # per-job start positions in read / text and direction
    qstart = torch.cat([qb[left] - 1, qe[right]])[order]          # left: walk down from qbeg-1
    tstart = torch.cat([rb[left] - 1, (rb + ln)[right]])[order]    # left: walk down from rbeg-1
    step = torch.where(side == 0, -torch.ones_like(side), torch.ones_like(side))
    qoff = torch.cumsum(qlen, 0) - qlen
    toff = torch.cumsum(tlen, 0) - tlen
    nq, nt = int(qlen.sum()), int(tlen.sum())
    jq = torch.repeat_interleave(torch.arange(qlen.numel(), device=dev), qlen)
    kq = torch.arange(nq, device=dev) - qoff[jq]
    qbases = reads.codes[jr[jq], qstart[jq] + step[jq] * kq]
    jt = torch.repeat_interleave(torch.arange(tlen.numel(), device=dev), tlen)
    kt = torch.arange(nt, device=dev) - toff[jt]
    tpos = tstart[jt] + step[jt] * kt
    rev = tpos >= l_pac
    fpos = torch.where(rev, 2 * l_pac - 1 - tpos, tpos)
    tb = genome_fwd[fpos]
    tbases = torch.where(rev, 3 - tb, tb)
    i32 = lambda x: x.to(torch.int32).contiguous()
    return ExtJobs(qbases.contiguous() if nq else torch.zeros(1, dtype=torch.uint8, device=dev), i32(qoff), i32(qlen),
                   tbases.contiguous() if nt else torch.zeros(1, dtype=torch.uint8, device=dev), i32(toff), i32(tlen),
                   i32(h0), i32(jr), i32(side))


def seeds_to_torch(s, n_reads: int, device) -> dict:
    """bmh_seeds_t (device pointers owned by the workspace) -> torch tensors (copies)."""
    from .lib import _memcpy_d2d
    ns = int(s.n_seeds)

    def rd(ptr, n, dt):
        t = torch.empty(max(n, 1), dtype=dt, device=device)
        if n:
            _memcpy_d2d(t.data_ptr(), ptr, n * t.element_size())
        return t[:n]
    return dict(rbeg=rd(s.d_rbeg, ns, torch.int64), qbeg=rd(s.d_qbeg, 2 * ns, torch.int32).view(-1, 2),
                score=rd(s.d_score, ns, torch.int32), n_ref_pos=rd(s.d_n_ref_pos, n_reads, torch.int32),
                prefix=rd(s.d_prefix, n_reads, torch.int32))
