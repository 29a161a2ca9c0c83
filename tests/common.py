"""Shared test helpers: seeded genomes/indexes/reads and extension-job generators."""
from __future__ import annotations

import functools
import os

import numpy as np

from bwamem_hip import fmindex, synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@functools.lru_cache(maxsize=8)
def genome_and_index(n_bases: int, seed: int = 42):
    g = synth.make_genome(n_bases, seed=seed)
    return g, fmindex.build_fmd_index(g)


def flat_reads(reads2d: np.ndarray):
    n, L = reads2d.shape
    return reads2d.reshape(-1), (np.arange(n, dtype=np.uint64) * L), np.full(n, L, np.uint32)


def ragged_reads(rows):
    lens = np.array([len(r) for r in rows], np.uint32)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    flat = np.concatenate([np.asarray(r, np.uint8) for r in rows]) if len(rows) and lens.sum() else np.zeros(0, np.uint8)
    return flat, offs, lens


def edge_reads(genome: np.ndarray, rng) -> list:
    """Reads the reference's tests would need: shorter than k, all N, N at ends, exact, revcomp,
    read reaching the genome start/end, homopolymer, tandem repeat."""
    n = genome.shape[0]
    rows = []
    rows.append(genome[100:110].copy())                         # shorter than min_seed_len
    rows.append(np.full(60, 4, np.uint8))                       # all N
    r = genome[500:650].copy(); r[0] = 4; r[-1] = 4; rows.append(r)
    rows.append(genome[1000:1150].copy())                       # exact forward
    rows.append(synth.revcomp(genome[2000:2150]))               # exact reverse
    rows.append(genome[0:150].copy())                           # touches text position 0
    rows.append(genome[n - 150:n].copy())                       # touches the fwd/rev boundary
    rows.append(synth.revcomp(genome[0:150]))                   # ends at the end of the text
    rows.append(np.zeros(150, np.uint8))                        # poly-A
    rows.append(np.tile(np.array([0, 1], np.uint8), 75))        # (AC)n
    r = genome[3000:3150].copy(); r[75] = 4; rows.append(r)     # N in the middle
    r = genome[4000:4019].copy(); rows.append(r)                # exactly min_seed_len
    r = genome[5000:5150].copy(); r[::20] = (r[::20] + 1) & 3; rows.append(r)  # mismatch every 20
    rows.append(rng.integers(0, 4, size=150).astype(np.uint8))  # unrelated
    rows.append(genome[6000:6001].copy())                       # single base
    return rows


def make_ext_jobs(n: int, rng, maxq: int = 281, allow_empty_query: bool = True):
    """Extension jobs with the structure chain2aln produces: a query that is a mutated copy of
    the start of the target, plus unrelated / homopolymer / N-containing cases."""
    qs, ts, h0 = [], [], []
    for _ in range(n):
        mode = int(rng.integers(0, 8))
        ql = int(rng.integers(0 if (mode == 7 and allow_empty_query) else 1, maxq + 1))
        tl = int(rng.integers(1, 2 * ql + 20))
        t = rng.integers(0, 4, size=tl).astype(np.uint8)
        if mode == 0:
            q = rng.integers(0, 4, size=ql).astype(np.uint8)
        elif mode == 1:
            b = int(rng.integers(0, 4))
            q = np.full(ql, b, np.uint8); t[:] = b
        else:
            out, j = [], 0
            while len(out) < ql:
                r = rng.random()
                if j >= tl:
                    out.append(int(rng.integers(0, 4))); continue
                if r < 0.02:
                    j += int(rng.integers(1, 4)); continue
                if r < 0.04:
                    out.extend(rng.integers(0, 4, size=int(rng.integers(1, 4))).tolist()); continue
                if r < 0.04 + 0.03 * mode:
                    out.append(int((t[j] + rng.integers(1, 4)) & 3)); j += 1; continue
                out.append(int(t[j])); j += 1
            q = np.array(out[:ql], np.uint8)
        if mode >= 5:
            q[rng.random(ql) < 0.02] = 4
            t[rng.random(tl) < 0.02] = 4
        qs.append(q); ts.append(t); h0.append(int(rng.integers(1, 200)))
    qlen = np.array([len(x) for x in qs], np.uint32); tlen = np.array([len(x) for x in ts], np.uint32)
    qoff = np.concatenate([[0], np.cumsum(qlen)[:-1]]).astype(np.uint32)
    toff = np.concatenate([[0], np.cumsum(tlen)[:-1]]).astype(np.uint32)
    q = np.concatenate(qs) if qlen.sum() else np.zeros(1, np.uint8)
    return q, qoff, qlen, np.concatenate(ts), toff, tlen, np.array(h0, np.uint32)


SEED_KEYS = ("rbeg", "qbeg", "score", "n_ref_pos", "prefix")


def assert_seeds_equal(a: dict, b: dict, what: str = ""):
    for k in SEED_KEYS:
        x, y = np.asarray(a[k]), np.asarray(b[k])
        assert x.shape == y.shape, f"{what}{k}: shape {x.shape} vs {y.shape}"
        if not np.array_equal(x, y):
            bad = np.nonzero(x.reshape(len(x), -1) != y.reshape(len(y), -1))[0][:5]
            raise AssertionError(f"{what}{k}: first mismatches at {bad}: {x[bad]} vs {y[bad]}")


def make_ext_jobs_fast(n: int, rng, maxq: int = 281):
    """Vectorised job generator for large differential runs (hundreds of thousands of jobs): the query is the start of the target
    with per-job substitution rates and one indel (a shift of the copy source behind a breakpoint), or unrelated; a few N."""
    qlen = rng.integers(1, maxq + 1, size=n).astype(np.uint32)
    tlen = (rng.integers(1, 2 * qlen.astype(np.int64) + 20)).astype(np.uint32)
    qoff = np.concatenate([[0], np.cumsum(qlen)[:-1]]).astype(np.uint32)
    toff = np.concatenate([[0], np.cumsum(tlen)[:-1]]).astype(np.uint32)
    t = rng.integers(0, 4, size=int(tlen.sum())).astype(np.uint8)
    job = np.repeat(np.arange(n), qlen)
    j = np.arange(int(qlen.sum())) - np.repeat(qoff.astype(np.int64), qlen)
    bp = rng.integers(0, qlen.astype(np.int64) + 1)
    shift = rng.integers(-3, 4, size=n) * (rng.random(n) < 0.4)
    src = j + np.where(j >= bp[job], shift[job], 0)
    ok = (src >= 0) & (src < tlen.astype(np.int64)[job])
    q = rng.integers(0, 4, size=j.shape[0]).astype(np.uint8)
    q[ok] = t[toff.astype(np.int64)[job[ok]] + src[ok]]
    rate = rng.choice([0.0, 0.01, 0.03, 0.08, 0.25, 1.0], size=n, p=[0.2, 0.25, 0.2, 0.15, 0.1, 0.1])
    m = rng.random(j.shape[0]) < rate[job]
    q[m] = (q[m] + rng.integers(1, 4, size=int(m.sum()))) & 3
    q[rng.random(q.shape[0]) < 0.002] = 4
    t[rng.random(t.shape[0]) < 0.002] = 4
    h0 = rng.integers(1, 200, size=n).astype(np.uint32)
    return q, qoff, qlen, t, toff, tlen, h0
