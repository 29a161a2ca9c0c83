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
               n_rate: float = 0.001):
    """Return (reads uint8 [n_reads, read_len] nt4 codes 0..4, truth dict).

    Reads are sampled at uniform positions from either strand; substitutions
    at sub_rate per base; indel_frac of the reads carry one 1-3 bp indel;
    n_rate of the bases become N.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    n = genome.shape[0]
    span = read_len + 4
    pos = rng.integers(0, n - span, size=n_reads)
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
               insert_sd: float = 35.0, sub_rate: float = 0.01, n_rate: float = 0.001):
    """Interleaved FR pairs (read 2i = first mate, 2i+1 = second mate): fragment of length ~N(insert_mean,
    insert_sd) at a uniform position, either strand; mate 1 = fragment start, mate 2 = reverse complement of
    the fragment end; substitutions / N as in make_reads.  Returns (reads uint8 [2*n_pairs, read_len], truth)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n = genome.shape[0]
    ins = np.clip(np.rint(rng.normal(insert_mean, insert_sd, size=n_pairs)).astype(np.int64), read_len, None)
    pos = rng.integers(0, n - ins.max() - 1, size=n_pairs)
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
