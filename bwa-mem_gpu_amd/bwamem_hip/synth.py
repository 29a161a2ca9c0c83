"""Seeded synthetic genomes and reads for the seed-and-extend hot path.

The reference ships no test data and there is no network,
so every genome/read set used by tests and bench.py is generated here from a
fixed seed (SURVEY.md section 8d "Synthetic inputs").

Genome: uniform ACGT plus planted repeat families (each copy diverged by
point substitutions, like real interspersed repeats).  Reads: uniform
position, either strand, per-base substitutions, a fraction of reads with
one short indel, a small fraction of N bases; written as single-line FASTA,
the only format the reference's seeding library parses
(/root/reference/src/GPUSeed/seed_gen.cu:1698-1728).
"""
from __future__ import annotations

import numpy as np

# nt4 codes used throughout BWA: A=0 C=1 G=2 T=3 N=4 (src/bwa.c nst_nt4_table)
_ASCII = np.frombuffer(b"ACGTN", dtype=np.uint8)


def make_genome(n_bases: int, seed: int = 42, repeat_frac: float = 0.10,
                repeat_len=(300, 6000), repeat_copies=(10, 2000),
                repeat_div: float = 0.08) -> np.ndarray:
    """Return a uint8 array of nt4 codes (0..3), length n_bases (no N)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    g = rng.integers(0, 4, size=n_bases, dtype=np.uint8)
    if repeat_frac <= 0 or n_bases < 4 * repeat_len[0]:
        return g
    budget = int(n_bases * repeat_frac)
    used = 0
    while used < budget:
        rl = int(rng.integers(repeat_len[0], min(repeat_len[1], max(repeat_len[0] + 1, n_bases // 8))))
        # log-uniform copy number
        lo, hi = np.log(repeat_copies[0]), np.log(repeat_copies[1])
        nc = int(np.exp(rng.uniform(lo, hi)))
        nc = max(2, min(nc, (budget - used) // rl + 2))
        cons = rng.integers(0, 4, size=rl, dtype=np.uint8)
        pos = rng.integers(0, n_bases - rl, size=nc)
        for p in pos:
            copy = cons.copy()
            nmut = rng.binomial(rl, repeat_div)
            if nmut:
                mp = rng.integers(0, rl, size=nmut)
                copy[mp] = (copy[mp] + rng.integers(1, 4, size=nmut, dtype=np.uint8)) & 3
            if rng.random() < 0.5:
                copy = (3 - copy)[::-1]
            g[p:p + rl] = copy
        used += rl * nc
    return g


def revcomp(codes: np.ndarray) -> np.ndarray:
    out = codes[::-1].copy()
    m = out < 4
    out[m] = 3 - out[m]
    return out


def make_reads(genome: np.ndarray, n_reads: int, read_len: int, seed: int = 7,
               sub_rate: float = 0.01, indel_frac: float = 0.05,
               n_rate: float = 0.001, holes=None):
    """Return (reads uint8 [n_reads, read_len] nt4 codes 0..4, truth dict).

    Reads are sampled at uniform positions from either strand; substitutions
    at sub_rate per base; indel_frac of the reads carry one 1-3 bp indel;
    n_rate of the bases become N.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    n = genome.shape[0]
    span = read_len + 4
    pos = rng.integers(0, n - span, size=n_reads) if not holes else sample_positions(rng, n, n_reads, span, holes)
    idx = pos[:, None] + np.arange(span)[None, :]
    frag = genome[idx]                                    # [n_reads, span]
    reads = frag[:, :read_len].copy()
    # indels
    has_indel = rng.random(n_reads) < indel_frac
    for r in np.nonzero(has_indel)[0]:
        k = int(rng.integers(1, 4))
        at = int(rng.integers(10, read_len - 10))
        if rng.random() < 0.5:                            # deletion from the read
            row = np.concatenate([frag[r, :at], frag[r, at + k:]])[:read_len]
        else:                                             # insertion into the read
            ins = rng.integers(0, 4, size=k, dtype=np.uint8)
            row = np.concatenate([frag[r, :at], ins, frag[r, at:]])[:read_len]
        reads[r] = row
    # substitutions
    sub = rng.random(reads.shape) < sub_rate
    reads[sub] = (reads[sub] + rng.integers(1, 4, size=int(sub.sum()), dtype=np.uint8)) & 3
    # strand
    rev = rng.random(n_reads) < 0.5
    rc = (3 - reads[rev])[:, ::-1]
    reads[rev] = rc
    # N bases
    nmask = rng.random(reads.shape) < n_rate
    reads[nmask] = 4
    return reads, {"pos": pos, "rev": rev, "has_indel": has_indel}


def make_pairs(genome: np.ndarray, n_pairs: int, read_len: int, seed: int = 7, insert_mean: float = 350.0,
               insert_sd: float = 35.0, sub_rate: float = 0.01, n_rate: float = 0.001, holes=None):
    """Interleaved FR pairs (read 2i = first mate, 2i+1 = second mate): fragment of length ~N(insert_mean,
    insert_sd) at a uniform position, either strand; mate 1 = fragment start, mate 2 = reverse complement of
    the fragment end; substitutions / N as in make_reads.  Returns (reads uint8 [2*n_pairs, read_len], truth)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n = genome.shape[0]
    ins = np.clip(np.rint(rng.normal(insert_mean, insert_sd, size=n_pairs)).astype(np.int64), read_len, None)
    pos = rng.integers(0, n - ins.max() - 1, size=n_pairs) if not holes else sample_positions(rng, n, n_pairs, int(ins.max()) + 1, holes)
    ar = np.arange(read_len)
    m1 = genome[pos[:, None] + ar[None, :]]
    m2f = genome[(pos + ins - read_len)[:, None] + ar[None, :]]
    m2 = (3 - m2f)[:, ::-1]
    flip = rng.random(n_pairs) < 0.5                       # fragment from the reverse strand: swap mate roles
    a = np.where(flip[:, None], m2, m1)
    b = np.where(flip[:, None], m1, m2)
    reads = np.empty((2 * n_pairs, read_len), np.uint8)
    reads[0::2] = a; reads[1::2] = b
    sub = rng.random(reads.shape) < sub_rate
    reads[sub] = (reads[sub] + rng.integers(1, 4, size=int(sub.sum()), dtype=np.uint8)) & 3
    reads[rng.random(reads.shape) < n_rate] = 4
    return reads, {"pos": pos, "insert": ins, "flip": flip}


def codes_to_ascii(codes: np.ndarray) -> np.ndarray:
    return _ASCII[codes]


def write_fasta_reads(path: str, reads: np.ndarray, prefix: str = "r") -> None:
    """Single-line FASTA, one '>' header line and one sequence line per read."""
    asc = codes_to_ascii(reads)
    with open(path, "wb") as f:
        for i in range(asc.shape[0]):
            f.write(b">%s%d\n" % (prefix.encode(), i))
            f.write(asc[i].tobytes())
            f.write(b"\n")


def write_fasta_genome(path: str, genome: np.ndarray, name: str = "chrS", width: int = 60) -> None:
    asc = codes_to_ascii(genome)
    with open(path, "wb") as f:
        f.write(b">" + name.encode() + b"\n")
        for i in range(0, asc.shape[0], width):
            f.write(asc[i:i + width].tobytes())
            f.write(b"\n")


# ---------------------------------------------------------------------------------------------------------------------
# hg38-scale synthetic genome, generated on the device (a 3.1 Gbp genome with planted repeats takes minutes in numpy).
# torch is plumbing here: device memory and a seeded random generator.

# lengths of chr1..22, X, Y of GRCh38 in Mbp: the proportions of the 24 contigs
_HG38_MBP = [248, 242, 198, 190, 181, 171, 159, 145, 138, 133, 135, 133, 114, 107, 102, 90, 83, 80, 58, 64, 46, 50, 156, 57]


def hg38_like_contigs(n_bases: int):
    """24 (name, length) pairs in the proportions of GRCh38's chromosomes, summing to n_bases."""
    tot = float(sum(_HG38_MBP))
    lens = [int(n_bases * m / tot) for m in _HG38_MBP]
    lens[0] += n_bases - sum(lens)
    names = [f"chr{i}" for i in range(1, 23)] + ["chrX", "chrY"]
    return list(zip(names, lens))


def make_genome_device(n_bases: int, device, seed: int = 42, repeat_frac: float = 0.5, return_meta: bool = False):
    """hg38-like synthetic genome as a uint8 tensor of nt4 codes 0..3 on `device`.

    Uniform ACGT, then planted on top of it, oldest first (later insertions overwrite earlier ones, like nested repeats):
      * 45 % of repeat_frac: mid-copy families (length 300-6000, 10-5000 copies, 3-25 % diverged from their consensus)
      * 30 %: one LINE-like family (6 kbp consensus, 5'-truncated copies, 2-20 % diverged)
      * 20 %: one SINE-like high-copy family (300 bp, ~1 M copies per 3.1 Gbp, 2-18 % diverged)
      * 2 %: tandem satellite arrays (171-bp and shorter monomers, arrays of 20-500 kbp, 8-20 % diverged per monomer)
      * 6 %: low-divergence segmental duplications (10-100 kbp of the genome itself copied 1-3 times at 0.5-3 %)
    and finally N-runs (telomeres, one centromere-sized gap and a few smaller gaps per contig, ~4 % of the genome) which --
    exactly as bns_fasta2bntseq does for the .pac (bwa_index/bntseq.c:249-251: `c = lrand48() & 3`) -- hold random
    bases; they are returned as `holes` so that reads are not drawn from them.  Copies are substitutions-only mutations of
    their source, half of them reverse-complemented.  Deterministic for a given (n_bases, seed) on a given device type."""
    import torch
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    rng = np.random.Generator(np.random.PCG64(seed))
    g = torch.randint(0, 4, (n_bases,), dtype=torch.uint8, device=dev, generator=gen)
    scale = n_bases / 3.1e9

    def put(p, src):
        # slice assignment through an elementwise kernel (codes are 0..3, so `& 3` is the identity): a contiguous
        # device-to-device copy_ goes through hipMemcpyAsync, which rocprofv3 --pmc crashes on (SIGSEGV inside the tool)
        torch.bitwise_and(src, 3, out=g[p:p + int(src.numel())])

    def plant(cons, n_copies, div_lo, div_hi, trunc=False, budget=None):
        """n_copies copies of cons (uint8 tensor) at distinct slots; returns bases planted"""
        rl = int(cons.numel())
        slots = n_bases // rl - 1
        n_copies = int(min(n_copies, slots // 2))
        if n_copies < 1:
            return 0
        done = 0
        CH = max(1, int(2e8) // rl)                       # copies per sub-batch: bounded temporaries
        slot_ids = torch.randperm(slots, device=dev, generator=gen)[:n_copies]
        for c0 in range(0, n_copies, CH):
            sl = slot_ids[c0:c0 + CH]
            k = int(sl.numel())
            if trunc:
                u = torch.rand(k, device=dev, generator=gen)
                ln = (rl * u * u).to(torch.int64).clamp_(min=min(100, rl))       # most copies are short 3' fragments
            else:
                ln = torch.full((k,), rl, dtype=torch.int64, device=dev)
            start = sl.to(torch.int64) * rl + (torch.rand(k, device=dev, generator=gen) * (rl - ln + 1).to(torch.float32)).to(torch.int64)
            off = torch.cumsum(ln, 0) - ln
            tot = int(ln.sum().item())
            cid = torch.repeat_interleave(torch.arange(k, device=dev), ln)
            j = torch.arange(tot, device=dev, dtype=torch.int64) - off[cid]
            flip = torch.rand(k, device=dev, generator=gen) < 0.5
            lc = ln[cid]
            fl = flip[cid]
            src = (rl - lc) + torch.where(fl, lc - 1 - j, j)                        # 3' end of the consensus is kept
            b = cons[src]
            b = torch.where(fl, 3 - b, b)
            div = div_lo + (div_hi - div_lo) * torch.rand(k, device=dev, generator=gen)
            mut = torch.rand(tot, device=dev, generator=gen) < div[cid]
            add = torch.randint(1, 4, (tot,), dtype=torch.uint8, device=dev, generator=gen)
            b = torch.where(mut, (b + add) & 3, b)
            g[start[cid] + j] = b
            done += tot
            del cid, j, lc, fl, src, b, mut, add
        return done

    budget = repeat_frac * n_bases
    planted = {}
    # mid-copy families
    want, used, nf = 0.45 * budget, 0, 0
    while used < want:
        rl = int(rng.integers(300, 6001))
        nc = int(np.exp(rng.uniform(np.log(10), np.log(5000))))
        nc = max(2, min(nc, int((want - used) // rl) + 2))
        dlo = float(rng.uniform(0.03, 0.15))
        cons = torch.randint(0, 4, (rl,), dtype=torch.uint8, device=dev, generator=gen)
        used += plant(cons, nc, dlo, dlo + 0.10)
        nf += 1
    planted["mid_families"] = (nf, used)
    cons = torch.randint(0, 4, (6000,), dtype=torch.uint8, device=dev, generator=gen)
    planted["line_like"] = plant(cons, int(0.30 * budget / 2000), 0.02, 0.20, trunc=True)      # mean truncated length = rl / 3
    cons = torch.randint(0, 4, (300,), dtype=torch.uint8, device=dev, generator=gen)
    planted["sine_like"] = plant(cons, int(0.20 * budget / 300), 0.02, 0.18)
    # tandem arrays
    want, used, na = 0.02 * budget, 0, 0
    while used < want:
        mono = int(rng.choice([171, 171, 68, 42, 5]))
        alen = int(rng.integers(20_000, 500_001) * min(1.0, max(scale, 0.02)))
        alen = max(mono * 4, min(alen, n_bases // 50))
        m = torch.randint(0, 4, (mono,), dtype=torch.uint8, device=dev, generator=gen)
        arr = m.repeat(alen // mono + 1)[:alen].clone()
        dv = float(rng.uniform(0.08, 0.20))
        mut = torch.rand(alen, device=dev, generator=gen) < dv
        arr = torch.where(mut, (arr + torch.randint(1, 4, (alen,), dtype=torch.uint8, device=dev, generator=gen)) & 3, arr)
        p = int(rng.integers(0, n_bases - alen))
        put(p, arr)
        used += alen; na += 1
    planted["tandem_arrays"] = (na, used)
    # segmental duplications of the genome itself
    want, used, nd = 0.06 * budget, 0, 0
    while used < want:
        ln = int(rng.integers(10_000, 100_001) * min(1.0, max(scale * 4, 0.02)))
        ln = max(1000, min(ln, n_bases // 100))
        src = int(rng.integers(0, n_bases - ln))
        seg = g[src:src + ln].clone()
        for _ in range(int(rng.integers(1, 4))):
            dv = float(rng.uniform(0.005, 0.03))
            mut = torch.rand(ln, device=dev, generator=gen) < dv
            c = torch.where(mut, (seg + torch.randint(1, 4, (ln,), dtype=torch.uint8, device=dev, generator=gen)) & 3, seg)
            if rng.random() < 0.5:
                c = (3 - c).flip(0)
            p = int(rng.integers(0, n_bases - ln))
            put(p, c)
            used += ln
        nd += 1
    planted["segdups"] = (nd, used)
    # N-runs (random bases in the .pac)
    contigs = hg38_like_contigs(n_bases)
    holes = []
    off = 0
    for _, cl in contigs:
        tel = max(10, int(10_000 * min(1.0, scale * 10)))
        cen = int(cl * 0.025)
        hs = [(off, tel), (off + cl - tel, tel), (off + int(cl * rng.uniform(0.3, 0.6)), cen)]
        for _ in range(4):
            hl = int(cl * rng.uniform(0.0005, 0.003))
            hs.append((off + int(rng.integers(tel, cl - tel - hl)), hl))
        for hb, hl in hs:
            if hl > 0:
                put(hb, torch.randint(0, 4, (hl,), dtype=torch.uint8, device=dev, generator=gen))
                holes.append((hb, hb + hl))
        off += cl
    holes.sort()
    if return_meta:
        return g, {"contigs": contigs, "holes": holes, "planted": planted}
    return g


def sample_positions(rng, n_genome: int, n: int, span: int, holes=None) -> np.ndarray:
    """n start positions of `span`-base windows, uniform over the genome outside the N-runs (`holes`: sorted (begin, end) pairs)"""
    pos = rng.integers(0, n_genome - span, size=n)
    if holes:
        hb = np.array([h[0] for h in holes], np.int64); he = np.array([h[1] for h in holes], np.int64)
        for _ in range(64):
            i = np.searchsorted(he, pos, side="right")              # first hole ending after pos
            i = np.minimum(i, len(hb) - 1)
            bad = (pos + span > hb[i]) & (pos < he[i])
            nb = int(bad.sum())
            if nb == 0:
                break
            pos[bad] = rng.integers(0, n_genome - span, size=nb)
    return pos
