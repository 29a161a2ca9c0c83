"""PCIe- and parse-inclusive rate of the reference-compatible seed_gpu() entry point (file in, host arrays out)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bwa-mem_gpu_amd"))
import numpy as np, torch
import bwamem_hip as B
from bwamem_hip import fmindex, synth
gsize = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200_000_000
nreads = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
work = "/tmp/seedfile"; os.makedirs(work, exist_ok=True)
g = synth.make_genome(gsize, seed=42)
idx = fmindex.build_fmd_index(g, device="cuda:0")
prefix = os.path.join(work, "g.fa"); fmindex.write_index(prefix, idx)
reads, _ = synth.make_reads(g, nreads, 150, seed=7)
fq = os.path.join(work, "r.fa"); synth.write_fasta_reads(fq, reads)
for it in range(2):
    t = time.time(); s = B.seed_file(prefix, fq, 19); dt = time.time() - t
    print("seed_file: %d reads, %d seeds in %.3f s -> %.2f Mreads/s (index load+upload, FASTA parse, H2D, kernels, D2H)" % (len(s["n_ref_pos"]), len(s["rbeg"]), dt, nreads / dt / 1e6), flush=True)
