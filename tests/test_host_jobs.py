"""CPU test of the host job builder (bmh_build_jobs is host code: no GPU needed): it reproduces the exact
multiset of extension jobs that the REFERENCE's own host code submitted for the same reads and seeds
(tests/golden/jobs_golden.npz, recorded on the MI355X box by scripts/make_jobs_golden.py through
BMH_GASAL_DUMP), and with the oracle as extension back-end the best region score of every read equals the
AS tag of the reference's SAM output."""
import hashlib
import os

import numpy as np

import common
from bwamem_hip import synth
from bwamem_hip.lib import HostJobs


def test_jobs_and_region_scores_match_reference_host_code(oracle):
    z = np.load(os.path.join(common.GOLDEN, "jobs_golden.npz"))
    g = synth.make_genome(int(z["n_genome"]), seed=int(z["genome_seed"]))
    reads = z["reads"]
    n, L = reads.shape
    seeds = {k: z[k] for k in ("rbeg", "qbeg", "score", "n_ref_pos", "prefix")}
    hj = HostJobs(g, reads.reshape(-1), np.arange(n, dtype=np.uint64) * L, np.full(n, L, np.uint32), seeds, n_threads=4)
    digs = []
    for i in range(hj.n_jobs):
        h0 = int(hj.h0[i])
        digs.append(hashlib.sha1(bytes([h0 & 255, h0 >> 8]) + hj.q[hj.qoff[i]:hj.qoff[i] + hj.qlen[i]].tobytes() + b"|" +
                                 hj.t[hj.toff[i]:hj.toff[i] + hj.tlen[i]].tobytes()).digest())
    digs.sort()
    want = [bytes(r) for r in z["job_digests"]]
    assert len(digs) == len(want) and digs == want
    # one thread or four: same batch
    hj1 = HostJobs(g, reads.reshape(-1), np.arange(n, dtype=np.uint64) * L, np.full(n, L, np.uint32), seeds, n_threads=1)
    assert hj1.n_jobs == hj.n_jobs and np.array_equal(hj1.q, hj.q) and np.array_equal(hj1.toff, hj.toff)
    # extension by the checker, region merge by the library: best score per read == AS of the reference's SAM
    out3, _, _ = oracle.extend_batch(*hj.jobs())
    regs = hj.merge(out3)
    best = np.full(n, -1, np.int64)
    np.maximum.at(best, regs[:, 0], regs[:, 1])
    as_tag = z["as_tag"]
    has = as_tag >= 0
    assert has.sum() > 0.9 * n and np.array_equal(best[has], as_tag[has])
    hj.free(); hj1.free()
