#!/usr/bin/env python3
"""Generates the golden vectors in tests/golden/ from the REFERENCE's own compiled C.

Runs only where /root/reference exists (the build container).  Everything expected here is
produced by reference code, none of it by our oracle or kernels:
  - ext_kat.npz    : reference ksw_extend2 (src/ksw.c:864, opt_ext=0) + decoy_cpu_align's rule
                     (src/bwamem.c:1893-1901) on seeded extension jobs, zdrop 0 and 100
  - seed_kat.npz   : reference bwt_smem1 / bwt_sa (src/bwt.c:563,105) on a 200 kbp seeded genome
                     (vanilla layout built in memory from the BWT symbols; SA computed by the
                     reference's bwt_cal_sa), 2000 seeded reads + edge-case reads
  - occ_kat.npz    : reference bwt_occ / bwt_sa at random rows incl. primary, 0, seq_len, block edges
  - ref_index_g20011.{fa,bwt,sa} : index files written by the reference's `bwa index` CLI
                     (bwa_index/, both passes of build_index.sh) for a 20011 bp seeded genome.
                     These three files must be produced separately with the two scratch builds of
                     bwa_index (OCC_INTV_SHIFT 7 for `-s sa`, 6 for `-s bwt`):
                        bwa_shift7 index -s sa -r 16 -p g g.fa ; bwa_shift6 index -s bwt -p g g.fa
                     pass --bwa7/--bwa6 to regenerate them.
"""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "bwa-mem_gpu_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import common  # noqa: E402
import oracle_py  # noqa: E402
from bwamem_hip import fmindex, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bwa7"); ap.add_argument("--bwa6")
    a = ap.parse_args()
    oracle_py.build(ref=True)
    ref = oracle_py.Ref()

    # ---- extension KATs
    jobs = common.make_ext_jobs(4000, np.random.default_rng(1234))
    out = {}
    for zd in (0, 100):
        o3, r6 = ref.extend_batch(*jobs, params=oracle_py.default_params(zdrop=zd))
        out[f"out3_z{zd}"] = o3; out[f"raw6_z{zd}"] = r6
    q, qoff, qlen, t, toff, tlen, h0 = jobs
    np.savez_compressed(os.path.join(HERE, "ext_kat.npz"), q=q, qoff=qoff, qlen=qlen, t=t, toff=toff, tlen=tlen, h0=h0, **out)

    # ---- seeding KATs
    g = synth.make_genome(200_000, seed=42)
    idx = fmindex.build_fmd_index(g)
    b = ref.bwt_from_index(idx)
    reads, _ = synth.make_reads(g, 2000, 150, seed=7)
    rows = [r for r in reads] + common.edge_reads(g, np.random.default_rng(3))
    flat, offs, lens = common.ragged_reads(rows)
    s = ref.seed_reads(b, flat, offs, lens, 19)
    np.savez_compressed(os.path.join(HERE, "seed_kat.npz"), genome=np.packbits(np.unpackbits(g[:, None], axis=1)[:, 6:].reshape(-1)),
                        n_genome=len(g), reads=flat, offs=offs, lens=lens, **{k: s[k] for k in s})

    # ---- occ / sa KATs
    rng = np.random.default_rng(99)
    ks = np.concatenate([rng.integers(0, idx.seq_len + 1, 6000), [0, 1, idx.primary - 1, idx.primary, idx.primary + 1,
                                                                 idx.seq_len, idx.seq_len - 1, 63, 64, 65, 127, 128]]).astype(np.uint64)
    occ = np.array([[ref.lib.ref_occ(b, int(k), c) for c in range(4)] for k in ks], np.uint64)
    sa = np.array([ref.lib.ref_sa(b, int(k)) for k in ks], np.uint64)
    np.savez_compressed(os.path.join(HERE, "occ_kat.npz"), k=ks, occ=occ, sa=sa)

    # ---- reference-built index files
    if a.bwa7 and a.bwa6:
        gg = synth.make_genome(20011, seed=42)
        with tempfile.TemporaryDirectory() as d:
            fa = os.path.join(d, "g.fa")
            synth.write_fasta_genome(fa, gg)
            subprocess.check_call([a.bwa7, "index", "-s", "sa", "-r", "16", "-p", os.path.join(d, "g"), fa], stderr=subprocess.DEVNULL)
            subprocess.check_call([a.bwa6, "index", "-s", "bwt", "-p", os.path.join(d, "g"), fa], stderr=subprocess.DEVNULL)
            shutil.copy(fa, os.path.join(HERE, "ref_index_g20011.fa"))
            shutil.copy(os.path.join(d, "g.bwt"), os.path.join(HERE, "ref_index_g20011.bwt"))
            shutil.copy(os.path.join(d, "g.sa"), os.path.join(HERE, "ref_index_g20011.sa"))
            shutil.copy(os.path.join(d, "g.pac"), os.path.join(HERE, "ref_index_g20011.pac"))
            shutil.copy(os.path.join(d, "g.ann"), os.path.join(HERE, "ref_index_g20011.ann"))
            shutil.copy(os.path.join(d, "g.amb"), os.path.join(HERE, "ref_index_g20011.amb"))
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
