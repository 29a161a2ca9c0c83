"""cProfile of one warm align_batch of bwamem_hip.aligner (1 Gbp genome, 1 M x 150 bp): where the host side of a batch spends its time"""
import os, sys, time, cProfile, pstats, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bwa-mem_gpu_amd"))
import numpy as np, torch
import bwamem_hip as B
from bwamem_hip.aligner import Aligner, read_fasta_reads
mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 1000
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
g = B.synth.make_genome(int(mbp * 1e6), seed=42)
idx = B.fmindex.build_fmd_index(g, device="cuda:0")
torch.cuda.empty_cache()
al = Aligner.from_memory(idx, g)
reads = B.synth.make_reads(g, n_reads, 150, seed=7)[0]
asc = B.synth.codes_to_ascii(reads)
fa = "/tmp/aligner_probe.fa"
with open(fa, "wb") as f:
    for i in range(0, len(asc), 200000):
        f.write(b"".join(b">r%08d\n" % j + asc[j].tobytes() + b"\n" for j in range(i, min(i + 200000, len(asc)))))
rs = read_fasta_reads(fa)
for _ in range(2):
    al.align_batch(rs, as_bytes="view")
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable(); txt = al.align_batch(rs, as_bytes="view"); pr.disable()
print("batch %.1f ms" % ((time.perf_counter() - t0) * 1e3))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:4500])
