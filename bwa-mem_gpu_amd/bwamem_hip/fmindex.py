"""FMD-index construction and the reference's on-disk GPU layout.

This is the "index converter/loader" row of SURVEY.md section 8f(2): it produces
exactly the files `build_index.sh` of the reference produces
(/root/reference/build_index.sh:46-68), so that the seeding library's loaders
(`bwt_restore_bwt_gpu`, `bwt_restore_sa_gpu`; reference
src/GPUSeed/seed_gen.cu:1386-1468) can read them:

  <prefix>.bwt : u64 primary, u64 L2[1..4], then per 64 BWT symbols one 32-byte
                 block {u32 occ[4]; u32 bwt[4]} (16 symbols per word, 2 bits,
                 MSB first) and one trailing occ quadruple
                 (writer: bwa_index/bwtindex.c:174-197).
  <prefix>.sa  : u64 primary, u64 L2[1..4], u64 sa_intv, u64 seq_len,
                 u32 sa[1..n_sa-1], u8 pack_size, u32 bits[pack_size*n_sa/32+1]
                 (writer: bwa_index/bwt.c:472-487; sampling :63-148).

The text indexed is forward strand followed by its reverse complement
(seq_len = 2*l_pac), as BWA does.  The suffix array is built by prefix doubling
with torch sorts, so the same code builds small test indexes on the CPU and
bench-scale indexes on the MI355X (torch is plumbing here, not the product).
"""
from __future__ import annotations

import struct
from dataclasses import dataclass

import numpy as np
import torch


@dataclass
class FMDIndex:
    primary: int
    L2: np.ndarray            # int64[5]
    seq_len: int
    bwt_words: np.ndarray     # uint32, interleaved occ/bwt blocks (GPU layout)
    sa_intv: int
    n_sa: int
    sa: np.ndarray            # uint32[n_sa], sa[0] = 0xFFFFFFFF
    sa_bits: np.ndarray       # uint32[pack_size*n_sa/32+1]
    pack_size: int


def _suffix_array(text: torch.Tensor) -> torch.Tensor:
    """Suffix array of text+$ (length n+1, $ smallest), int64, by prefix doubling."""
    dev = text.device
    n = int(text.numel())
    n1 = n + 1
    K = 12
    # base-5 digits: 0 = past the end / sentinel, 1..4 = A..T
    d = torch.zeros(n1 + K, dtype=torch.int64, device=dev)
    d[:n] = text.to(torch.int64) + 1
    key = torch.zeros(n1, dtype=torch.int64, device=dev)
    for j in range(K):
        key = key * 5 + d[j:j + n1]
    del d
    order = torch.argsort(key)
    sk = key[order]
    del key
    flag = torch.ones(n1, dtype=torch.int64, device=dev)
    flag[1:] = (sk[1:] != sk[:-1]).to(torch.int64)
    del sk
    r = torch.cumsum(flag, 0)                 # 1-based dense rank in sorted order
    nranks = int(r[-1])
    rank = torch.empty(n1, dtype=torch.int64, device=dev)
    rank[order] = r
    h = K
    mult = n1 + 1
    if mult * mult >= 2 ** 63:
        raise ValueError("text too long for 64-bit pair keys")
    while nranks < n1:
        r2 = torch.zeros(n1, dtype=torch.int64, device=dev)
        if h < n1:
            r2[:n1 - h] = rank[h:]
        key = rank * mult + r2
        del r2
        order = torch.argsort(key)
        sk = key[order]
        del key
        flag[0] = 1
        flag[1:] = (sk[1:] != sk[:-1]).to(torch.int64)
        del sk
        r = torch.cumsum(flag, 0)
        nranks = int(r[-1])
        rank[order] = r
        h *= 2
    return order


def build_fmd_index(genome_fwd: np.ndarray, sa_intv: int = 16, device: str | None = None) -> FMDIndex:
    """Build the FMD index of genome_fwd (nt4 codes 0..3) + its reverse complement."""
    assert genome_fwd.dtype == np.uint8 and genome_fwd.max(initial=0) < 4
    dev = torch.device(device) if device else torch.device("cpu")
    fwd = torch.from_numpy(np.ascontiguousarray(genome_fwd)).to(dev)
    text = torch.cat([fwd, (3 - fwd).flip(0)])
    n = int(text.numel())
    SA = _suffix_array(text)                                  # n+1 rows
    primary = int(torch.nonzero(SA == 0)[0, 0])
    # BWT column with the $ row removed (is.c is_bwt semantics)
    prev = torch.where(SA > 0, SA - 1, torch.zeros_like(SA))
    B = text[prev]
    keep = torch.ones(n + 1, dtype=torch.bool, device=dev)
    keep[primary] = False
    bwt = B[keep]                                             # length n, codes 0..3
    del B, keep, prev
    cnt = torch.bincount(text.to(torch.int64), minlength=4).cpu().numpy()
    L2 = np.zeros(5, dtype=np.int64)
    L2[1:] = np.cumsum(cnt)
    # SA samples: rows isa % intv == 0 of the (n+1)-row matrix
    n_sa = (n + sa_intv) // sa_intv
    sa_full = SA[::sa_intv][:n_sa].cpu().numpy().astype(np.int64)
    del SA
    assert sa_full.shape[0] == n_sa
    sa32 = (sa_full & 0xFFFFFFFF).astype(np.uint32)
    hi = (sa_full >> 32).astype(np.uint32)
    msb = int(n >> 32).bit_length()
    if msb <= 1:
        pack_size, pack_mask = 1, (1 if msb == 1 else 0)
    else:
        raise ValueError("seq_len >= 2^33 not supported by the reference's GPU layout")
    bits = np.zeros(pack_size * n_sa // 32 + 1, dtype=np.uint32)
    idx = np.nonzero(hi & pack_mask)[0]
    np.bitwise_or.at(bits, idx // 32, (np.uint32(1) << (idx % 32).astype(np.uint32)))
    sa32[0] = 0xFFFFFFFF
    bits[0] |= np.uint32(pack_mask)
    # pack BWT, 16 symbols per word, MSB first, zero padded
    n16 = (n + 15) // 16
    pad = torch.zeros(n16 * 16, dtype=torch.int64, device=dev)
    pad[:n] = bwt.to(torch.int64)
    sh = torch.arange(15, -1, -1, device=dev, dtype=torch.int64) * 2
    words = (pad.view(n16, 16) << sh).sum(1)                  # < 2^32
    nblk = (n + 63) // 64
    # occ before each block (+ the trailing total)
    pad64 = torch.full((nblk * 64,), 4, dtype=torch.int64, device=dev)
    pad64[:n] = bwt.to(torch.int64)
    blk = pad64.view(nblk, 64)
    occ = torch.zeros(nblk + 1, 4, dtype=torch.int64, device=dev)
    for c in range(4):
        occ[1:, c] = torch.cumsum((blk == c).sum(1), 0)
    wpad = torch.zeros(nblk * 4, dtype=torch.int64, device=dev)
    wpad[:n16] = words
    out = torch.zeros(nblk, 8, dtype=torch.int64, device=dev)
    out[:, :4] = occ[:nblk]
    out[:, 4:] = wpad.view(nblk, 4)
    flat = out.view(-1)
    # the writer emits words only for i < seq_len (i%16==0), so a partial last block is short
    n_words_last = n16 - (nblk - 1) * 4
    flat = flat[: (nblk - 1) * 8 + 4 + n_words_last]
    flat = torch.cat([flat, occ[nblk]])
    bwt_words = flat.cpu().numpy().astype(np.uint32)
    return FMDIndex(primary=primary, L2=L2, seq_len=n, bwt_words=bwt_words, sa_intv=sa_intv,
                    n_sa=n_sa, sa=sa32, sa_bits=bits, pack_size=pack_size)


@dataclass
class DeviceFMDIndex:
    """What bmh_index_build leaves in HBM: the arguments of bmh_index_from_device (torch tensors are plumbing: device memory)."""
    primary: int
    L2: np.ndarray
    seq_len: int
    bwt_t: "torch.Tensor"      # int32 [(ceil(seq_len/64)+1)*8]: whole 32-byte blocks, the last one holds the totals
    sa_intv: int
    sa_t: "torch.Tensor"       # int32 [n_sa]
    bits_t: "torch.Tensor"     # int32 [n_sa/32+1]
    stats: dict


def pack_pac_device(codes_t: torch.Tensor) -> torch.Tensor:
    """nt4 codes 0..3 (uint8, on the device) -> the .pac body: 4 bases per byte, first base in the top bits; padded by 64 bytes
    (the kernels fetch the text in aligned words)."""
    n = int(codes_t.numel())
    pad = (-n) % 4
    if pad:
        codes_t = torch.cat([codes_t, torch.zeros(pad, dtype=torch.uint8, device=codes_t.device)])
    v = codes_t.view(-1, 4)
    pac = (v[:, 0] << 6) | (v[:, 1] << 4) | (v[:, 2] << 2) | v[:, 3]
    return torch.cat([pac, torch.zeros(64, dtype=torch.uint8, device=codes_t.device)]).contiguous()


def unpack_pac_device(pac_t: torch.Tensor, l_pac: int) -> torch.Tensor:
    """the .pac body on the device -> nt4 codes 0..3 (uint8 [l_pac])"""
    nb = (l_pac + 3) // 4
    b = pac_t[:nb]
    return torch.stack([(b >> 6) & 3, (b >> 4) & 3, (b >> 2) & 3, b & 3], dim=1).reshape(-1)[:l_pac].contiguous()


def build_fmd_index_device(pac_t: torch.Tensor, l_pac: int, sa_intv: int = 1, verify: bool = False) -> DeviceFMDIndex:
    """bmh_index_build: the FMD index of fwd . revcomp(fwd) built by the HIP library in HBM (hg38 scale: seq_len up to 2^33)."""
    from .lib import BuildStats, _err, _u64p, load_library
    import ctypes as C
    L = load_library()
    dev = pac_t.device
    n = 2 * int(l_pac)
    nblk = (n + 63) // 64
    n_sa = (n + sa_intv) // sa_intv
    bwt_t = torch.empty((nblk + 1) * 8, dtype=torch.int32, device=dev)
    sa_t = torch.empty(n_sa, dtype=torch.int32, device=dev)
    bits_t = torch.empty(n_sa // 32 + 1, dtype=torch.int32, device=dev)
    primary = C.c_uint64(0)
    L2 = np.zeros(5, dtype=np.uint64)
    st = BuildStats()
    torch.cuda.synchronize(dev)
    rc = L.bmh_index_build(pac_t.data_ptr(), l_pac, sa_intv, bwt_t.data_ptr(), sa_t.data_ptr(), bits_t.data_ptr(), C.byref(primary),
                           L2.ctypes.data_as(_u64p), 1 if verify else 0, C.byref(st))
    if rc != 0:
        raise RuntimeError(f"bmh_index_build rc={rc}: " + _err(L))
    stats = {k: getattr(st, k) for k, _ in BuildStats._fields_}
    return DeviceFMDIndex(int(primary.value), L2.astype(np.int64), n, bwt_t, sa_intv, sa_t, bits_t, stats)


def host_index_to_device_form(idx: FMDIndex, device) -> DeviceFMDIndex:
    """FMDIndex (file layout, host) -> tensors on `device` in the form bmh_index_from_device takes (blocks padded to whole units)"""
    nblk = (idx.seq_len + 63) // 64 + 1
    bwt = torch.zeros(nblk * 8, dtype=torch.int32, device=device)
    n16 = (idx.seq_len + 15) // 16
    body = (nblk - 2) * 8 + 4 + (n16 - (nblk - 2) * 4)          # words before the trailing totals
    w = torch.from_numpy(idx.bwt_words.view(np.int32).copy())
    bwt[:body] = w[:body].to(device)
    bwt[(nblk - 1) * 8:(nblk - 1) * 8 + 4] = w[body:body + 4].to(device)
    sa = torch.from_numpy(idx.sa.view(np.int32).copy()).to(device)
    bits = torch.from_numpy(idx.sa_bits.view(np.int32).copy()).to(device)
    return DeviceFMDIndex(idx.primary, np.asarray(idx.L2, dtype=np.int64), idx.seq_len, bwt, idx.sa_intv, sa, bits, {})


def device_index_to_host(d: DeviceFMDIndex, sa_intv: int = 16) -> FMDIndex:
    """Host copy in the reference's file layout (what write_index writes and the oracle reads), with the samples of every
    sa_intv-th row (a multiple of the device index's interval)."""
    n = d.seq_len
    nblk = (n + 63) // 64
    n16 = (n + 15) // 16
    n_words_last = n16 - (nblk - 1) * 4
    w = d.bwt_t[: (nblk - 1) * 8 + 4 + n_words_last].cpu().numpy().view(np.uint32)
    tot = (d.L2[1:] - d.L2[:-1]).astype(np.uint32)
    words = np.concatenate([w, tot])
    assert sa_intv % d.sa_intv == 0
    step = sa_intv // d.sa_intv
    n_sa = (n + sa_intv) // sa_intv
    sa = d.sa_t[::step][:n_sa].cpu().numpy().view(np.uint32).copy()
    assert sa.shape[0] == n_sa
    bits = np.zeros(n_sa // 32 + 1, dtype=np.uint32)
    if n >> 32:
        rows = torch.arange(0, n_sa, device=d.sa_t.device, dtype=torch.int64) * step
        hb = ((d.bits_t[rows >> 5] >> (rows & 31).to(torch.int32)) & 1).cpu().numpy().astype(bool)
        idx = np.nonzero(hb)[0]
        np.bitwise_or.at(bits, idx // 32, (np.uint32(1) << (idx % 32).astype(np.uint32)))
    sa[0] = 0xFFFFFFFF
    return FMDIndex(primary=d.primary, L2=d.L2.copy(), seq_len=n, bwt_words=words, sa_intv=sa_intv, n_sa=n_sa, sa=sa, sa_bits=bits, pack_size=1)


def write_index(prefix: str, idx: FMDIndex) -> None:
    with open(prefix + ".bwt", "wb") as f:
        f.write(struct.pack("<Q", idx.primary))
        f.write(struct.pack("<4Q", *[int(x) for x in idx.L2[1:]]))
        f.write(idx.bwt_words.astype("<u4").tobytes())
    with open(prefix + ".sa", "wb") as f:
        f.write(struct.pack("<Q", idx.primary))
        f.write(struct.pack("<4Q", *[int(x) for x in idx.L2[1:]]))
        f.write(struct.pack("<Q", idx.sa_intv))
        f.write(struct.pack("<Q", idx.seq_len))
        f.write(idx.sa[1:].astype("<u4").tobytes())
        f.write(struct.pack("<B", idx.pack_size))
        f.write(idx.sa_bits.astype("<u4").tobytes())


def read_index(prefix: str) -> FMDIndex:
    """Python mirror of bwt_restore_bwt_gpu / bwt_restore_sa_gpu (seed_gen.cu:1386-1468)."""
    raw = np.fromfile(prefix + ".bwt", dtype=np.uint8)
    hdr = np.frombuffer(raw[:40].tobytes(), dtype="<u8")
    primary = int(hdr[0])
    L2 = np.zeros(5, dtype=np.int64)
    L2[1:] = hdr[1:5].astype(np.int64)
    words = np.frombuffer(raw[40:].tobytes(), dtype="<u4").copy()
    seq_len = int(L2[4])
    with open(prefix + ".sa", "rb") as f:
        h = np.frombuffer(f.read(56), dtype="<u8")
        assert int(h[0]) == primary and int(h[6]) == seq_len
        sa_intv = int(h[5])
        n_sa = (seq_len + sa_intv) // sa_intv
        sa = np.empty(n_sa, dtype=np.uint32)
        sa[0] = 0xFFFFFFFF
        sa[1:] = np.frombuffer(f.read(4 * (n_sa - 1)), dtype="<u4")
        pack_size = struct.unpack("<B", f.read(1))[0]
        bits = np.frombuffer(f.read(4 * (pack_size * n_sa // 32 + 1)), dtype="<u4").copy()
        bits[0] |= 1
    return FMDIndex(primary, L2, seq_len, words, sa_intv, n_sa, sa, bits, pack_size)


def write_bns(prefix: str, genome_fwd: np.ndarray, name: str = "chrS", contigs=None) -> None:
    """.pac / .ann / .amb of N-free sequences, byte-identical to what the reference's `bwa index` writes
    (bwa_index/bntseq.c:66-95 bns_dump, :300-326 pac tail).  contigs: optional list of (name, length) whose lengths
    sum to len(genome_fwd) -- the concatenation is the packed reference; default one sequence `name`."""
    l_pac = int(genome_fwd.shape[0])
    if contigs is None:
        contigs = [(name, l_pac)]
    assert sum(int(c[1]) for c in contigs) == l_pac
    pad = (-l_pac) % 4
    codes = np.concatenate([genome_fwd, np.zeros(pad, np.uint8)]).reshape(-1, 4)
    pac = ((codes[:, 0] << 6) | (codes[:, 1] << 4) | (codes[:, 2] << 2) | codes[:, 3]).astype(np.uint8)
    with open(prefix + ".pac", "wb") as f:
        f.write(pac.tobytes())
        if l_pac % 4 == 0:
            f.write(b"\x00")
        f.write(bytes([l_pac % 4]))
    with open(prefix + ".ann", "w") as f:
        f.write(f"{l_pac} {len(contigs)} 11\n")
        off = 0
        for cname, clen in contigs:
            f.write(f"0 {cname} (null)\n{off} {int(clen)} 0\n")
            off += int(clen)
    with open(prefix + ".amb", "w") as f:
        f.write(f"{l_pac} {len(contigs)} 0\n")
