"""Direct pins of the HIP path to the REFERENCE: the known-answer vectors the reference's own compiled C produced
(tests/golden/{ext,seed,occ}_kat.npz, tests/golden/make_golden.py) go straight into the C ABI of libbwamem_hip.so, and -- where
oracle/_ref/libref.so travelled to this box -- the reference's ksw_extend2 / bwt_smem1 / bwt_sa are run beside the kernels on fresh
inputs.  No restatement in between (the oracle-vs-kernel tests are tests/test_gpu_parity.py)."""
import os

import numpy as np
import pytest

import common
from test_gpu_parity import gpu_extend, gpu_seed, hip  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu
G = common.GOLDEN


def _golden_genome_index():
    from bwamem_hip import fmindex
    z = np.load(os.path.join(G, "seed_kat.npz"))
    n = int(z["n_genome"])
    bits = np.unpackbits(z["genome"])[: 2 * n].reshape(n, 2)
    g = (bits[:, 0] * 2 + bits[:, 1]).astype(np.uint8)
    return g, fmindex.build_fmd_index(g), z


@pytest.mark.parametrize("packed", [1, 0])
@pytest.mark.parametrize("zdrop", [0, 100])
def test_extension_reproduces_reference_vectors(hip, zdrop, packed):
    """ext_kat.npz (4 000 jobs through the reference's ksw_extend2 + the local/to-end rule, src/ksw.c:864-986, src/bwamem.c:1893-1901)
    -> bmh_extend_batch: raw 6-tuples and the three GASAL2 results, packed 16-bit and 32-bit kernels."""
    z = np.load(os.path.join(G, "ext_kat.npz"))
    jobs = tuple(z[k] for k in ("q", "qoff", "qlen", "t", "toff", "tlen", "h0"))
    got3, got6 = gpu_extend(hip, jobs, zdrop=zdrop, packed=packed)
    assert np.array_equal(got6, z[f"raw6_z{zdrop}"])
    assert np.array_equal(got3, z[f"out3_z{zdrop}"])


@pytest.mark.parametrize("with_text", [False, True])
def test_seeding_reproduces_reference_vectors(hip, with_text):
    """seed_kat.npz (2 015 reads through the reference's bwt_smem1 + bwt_sa, src/bwt.c:483-566,105-115) -> bmh_seed_batch: the whole
    mem_seed_v_gpu; with and without the 2-bit text resident (the unique-interval shortcut)."""
    g, idx, z = _golden_genome_index()
    got = gpu_seed(hip, idx, z["reads"], z["offs"], z["lens"], 19, genome=g if with_text else None, densify=1 if with_text else None)
    want = {k: z[k] for k in common.SEED_KEYS}
    common.assert_seeds_equal(got, want)
    assert got["n_smems"] == len(z["smem_k"])


def test_rank_primitives_reproduce_reference_vectors(hip):
    """occ_kat.npz (bwt_occ for the four symbols and bwt_sa at 6 012 rows incl. 0, primary, seq_len and block boundaries) ->
    bmh_index_probe on the native bit-plane blocks; LF through SA[LF(k)] = SA[k] - 1."""
    g, idx, _ = _golden_genome_index()
    z = np.load(os.path.join(G, "occ_kat.npz"))
    index = hip.Index.upload(idx)
    try:
        k = z["k"].astype(np.uint64)
        assert np.array_equal(index.probe(k, "occ4"), z["occ"])
        assert np.array_equal(index.probe(k, "sa"), z["sa"])
        inner = k[(k <= idx.seq_len)]
        lf = index.probe(inner, "lf")
        sa_k, sa_lf = index.probe(inner, "sa"), index.probe(lf, "sa")
        n = np.uint64(idx.seq_len)
        # SA[LF(k)] = SA[k] - 1; row 0 is the sentinel's suffix (position n, which the reference stores as -1), row `primary`
        # (SA = 0) steps to row 0
        with np.errstate(over="ignore"):
            assert np.array_equal(sa_lf, np.where(inner == 0, n - np.uint64(1), sa_k - np.uint64(1)))
        # denser samples change nothing
        index.densify_sa(1)
        assert np.array_equal(index.probe(k, "sa"), z["sa"])
        # rows -1 and seq_len of the four-symbol form (bwt_2occ4's k - 1 at k = 0)
        edge = index.probe(np.array([np.uint64(2**64 - 1), np.uint64(idx.seq_len)], np.uint64), "occ4")
        assert np.array_equal(edge[0], np.zeros(4, np.uint64))
        assert np.array_equal(edge[1], np.diff(np.asarray(idx.L2, np.uint64)))
    finally:
        index.free()


def test_kernels_beside_the_reference_code(hip, ref):
    """HIP vs oracle/_ref/libref.so (the reference's own ksw.c / bwt.c compiled with plain gcc) on fresh seeded inputs: 200 000
    extension jobs (two z-drop settings) and 6 000 reads incl. the edge-case reads."""
    import oracle_py
    rng = np.random.default_rng(2024)
    jobs = common.make_ext_jobs_fast(200_000, rng, maxq=281)
    for zd in (0, 100):
        want3, want6 = ref.extend_batch(*jobs, params=oracle_py.default_params(zdrop=zd))
        got3, got6 = gpu_extend(hip, jobs, zdrop=zd)
        bad = np.flatnonzero((got6 != want6).any(1) | (got3 != want3).any(1))
        assert bad.size == 0, f"z-drop {zd}: {bad.size} jobs differ from the reference, first {bad[:3]}: {got6[bad[:3]]} vs {want6[bad[:3]]}"
    g, idx = common.genome_and_index(400_000, seed=31)
    b = ref.bwt_from_index(idx)
    reads, _ = hip.synth.make_reads(g, 6000, 150, seed=32, sub_rate=0.02)
    rows = [r for r in reads] + common.edge_reads(g, np.random.default_rng(33))
    flat, offs, lens = common.ragged_reads(rows)
    want = ref.seed_reads(b, flat, offs, lens, 19)
    common.assert_seeds_equal(gpu_seed(hip, idx, flat, offs, lens), want, what="rank walk: ")
    common.assert_seeds_equal(gpu_seed(hip, idx, flat, offs, lens, genome=g, densify=1), want, what="text shortcut: ")
