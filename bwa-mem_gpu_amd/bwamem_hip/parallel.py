"""Multi-GPU plumbing: read sharding and the one-off index broadcast.

The hot path shards embarrassingly: reads are independent units and the only
shared state is the read-only FMD index.  One process per
GPU; rank r takes the contiguous read range shard_range(n, r, world), so the
per-read output order and the prefix sums rebase trivially; rank 0 owns the
index and broadcasts it ONCE over RCCL/xGMI (torch.distributed backend "nccl"
on ROCm) before any batch is processed.  There is no collective on the data path.
The same code runs on the gloo backend with CPU tensors (tests/test_parallel.py).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int, multiple: int = 1):
    """Contiguous shard [lo, hi) of rank; boundaries are multiples of `multiple`
    (2 keeps interleaved read pairs together)."""
    units = n_items // multiple
    lo = units * rank // world * multiple
    hi = units * (rank + 1) // world * multiple
    if rank == world - 1:
        hi = n_items
    return lo, hi


def rebase_prefix(prefix_per_shard, n_ref_per_shard):
    """Concatenate per-shard mem_seed_v_gpu prefix sums into file-global ones."""
    out, base = [], 0
    for p, n in zip(prefix_per_shard, n_ref_per_shard):
        out.append(np.asarray(p, dtype=np.uint64) + base)
        base += int(np.asarray(n, dtype=np.uint64).sum())
    return np.concatenate(out).astype(np.uint32) if out else np.zeros(0, np.uint32)


def broadcast_index(idx, device, src: int = 0, world: int | None = None):
    """rank `src` passes an fmindex.FMDIndex, the others None.  Returns
    (header dict, bwt words, sa, sa_bits) as int32 tensors on `device`, the bwt
    tensor padded to whole 32-byte blocks (+1) as bmh_index_from_device requires."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    use_dist = dist.is_initialized()
    rank = dist.get_rank() if use_dist else 0
    hdr = torch.zeros(12, dtype=torch.int64)
    if rank == src:
        hdr[0] = idx.primary
        hdr[1:6] = torch.from_numpy(np.asarray(idx.L2, dtype=np.int64))
        hdr[6] = idx.seq_len; hdr[7] = idx.sa_intv; hdr[8] = idx.n_sa
        hdr[9] = idx.bwt_words.shape[0]; hdr[10] = idx.sa_bits.shape[0]
    if use_dist:
        h = hdr.to(device)
        dist.broadcast(h, src)
        hdr = h.cpu()
    seq_len, n_sa, n_bits = int(hdr[6]), int(hdr[8]), int(hdr[10])
    nblk = (seq_len + 63) // 64 + 1
    bwt = torch.zeros(nblk * 8, dtype=torch.int32, device=device)
    sa = torch.zeros(n_sa, dtype=torch.int32, device=device)
    bits = torch.zeros(n_bits, dtype=torch.int32, device=device)
    if rank == src:
        w = torch.from_numpy(idx.bwt_words.view(np.int32))
        bwt[: w.numel()] = w.to(device)
        sa.copy_(torch.from_numpy(idx.sa.view(np.int32)))
        bits.copy_(torch.from_numpy(idx.sa_bits.view(np.int32)))
    if use_dist:
        # three large point-to-multipoint transfers, issued back to back (one ring/tree each)
        dist.broadcast(bwt, src)
        dist.broadcast(sa, src)
        dist.broadcast(bits, src)
    header = dict(primary=int(hdr[0]), L2=hdr[1:6].numpy().astype(np.uint64), seq_len=seq_len, sa_intv=int(hdr[7]),
                  n_sa=n_sa)
    return header, bwt, sa, bits


def _bcast_chunked(t: torch.Tensor, src: int, chunk: int = 1 << 28) -> None:
    """one dist.broadcast per <= chunk elements (a dense hg38 suffix array is 6.2e9 words; transfers stay a few GB each)"""
    flat = t.view(-1)
    for o in range(0, flat.numel(), chunk):
        dist.broadcast(flat[o:o + chunk], src)


def broadcast_built_index(d, pac_t, meta, device, src: int = 0, with_text: bool = True):
    """The index bmh_index_build left in rank `src`'s HBM (fmindex.DeviceFMDIndex + the 2-bit pac + the genome's metadata) to
    every rank: `src` passes them, the others pass None.  One header, one object broadcast (metadata) and the four arrays
    (pac, Occ/BWT blocks, suffix array, its high bits) over RCCL/xGMI -- the only collective of the run, outside the data path.
    Returns (DeviceFMDIndex, pac_t, meta) on every rank.  Without an initialised process group it returns its arguments.
    with_text=False: every rank already holds the 2-bit text and the metadata (it generated the same genome itself while `src`
    built the index); only the header and the three index arrays travel, pac_t / meta come back as they were passed."""
    from .fmindex import DeviceFMDIndex
    if not dist.is_initialized():
        return d, pac_t, meta
    rank = dist.get_rank()
    hdr = torch.zeros(16, dtype=torch.int64, device=device)
    if rank == src:
        vals = [d.primary, *[int(x) for x in d.L2], d.seq_len, d.sa_intv, d.bwt_t.numel(), d.sa_t.numel(), d.bits_t.numel(), pac_t.numel()]
        hdr[: len(vals)] = torch.tensor(vals, dtype=torch.int64)
    dist.broadcast(hdr, src)
    h = hdr.cpu().tolist()
    box = [meta]
    if with_text:
        box = [meta if rank == src else None]
        dist.broadcast_object_list(box, src=src, device=device if device.type != "cpu" else None)
    if rank != src:
        d = DeviceFMDIndex(primary=h[0], L2=np.array(h[1:6], dtype=np.int64), seq_len=h[6], bwt_t=torch.empty(h[8], dtype=torch.int32, device=device),
                           sa_intv=h[7], sa_t=torch.empty(h[9], dtype=torch.int32, device=device),
                           bits_t=torch.empty(h[10], dtype=torch.int32, device=device), stats={})
        if with_text:
            pac_t = torch.empty(h[11], dtype=torch.uint8, device=device)
    for t in ((pac_t,) if with_text else ()) + (d.bwt_t, d.sa_t, d.bits_t):
        _bcast_chunked(t, src)
    return d, pac_t, box[0]
