"""End-to-end wall clock of bwamem_hip.aligner on the bench workload: ASCII reads in host memory -> SAM text in host memory
(H2D, every device stage, the host tail, D2H, text formatting).  usage: aligner_probe.py [genome_mbp] [n_reads] [pe]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bwa-mem_gpu_amd"))
import numpy as np, torch
import bwamem_hip as B
from bwamem_hip.aligner import Aligner
mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 1000
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
pe = len(sys.argv) > 3 and sys.argv[3] == "pe"
g = B.synth.make_genome(int(mbp * 1e6), seed=42)
idx = B.fmindex.build_fmd_index(g, device="cuda:0")
torch.cuda.empty_cache()
al = Aligner.from_memory(idx, g)
reads = (B.synth.make_pairs(g, n_reads // 2, 150, seed=7) if pe else B.synth.make_reads(g, n_reads, 150, seed=7))[0]
fa = "/tmp/aligner_probe.fa"
asc = B.synth.codes_to_ascii(reads)
with open(fa, "wb") as f:
    rec = np.empty((len(asc), 152), np.uint8)                 # fixed-width records: >r0000000\n + bases + \n, written in one go
    names = np.char.zfill(np.arange(len(asc) // (2 if pe else 1)).astype(str), 8)
    for i in range(0, len(asc), 200000):
        blk = asc[i:i + 200000]
        lines = [b">" + (b"p" if pe else b"r") + names[(j // 2) if pe else j].encode() + b"\n" + blk[j - i].tobytes() + b"\n" for j in range(i, i + len(blk))]
        f.write(b"".join(lines))
from bwamem_hip.aligner import read_fasta_reads
for it in range(3):
    t0 = time.perf_counter()
    rs = read_fasta_reads(fa)
    t1 = time.perf_counter()
    txt = al.align_batch(rs, id0=0, paired=pe, as_bytes="view")
    dt = time.perf_counter() - t1
    print("%s: %d reads: FASTA parse %.1f ms; reads in host memory -> %d bytes of SAM in %.1f ms = %.2f Mreads/s end to end" %
          ("PE" if pe else "SE", len(rs), (t1 - t0) * 1e3, len(txt), dt * 1e3, len(rs) / dt / 1e6), flush=True)
# the whole file through align_file in four batches (single-end: the two-stage pipeline; paired: one batch after the other)
class _Sink:
    mode = "wb"
    def __init__(self): self.n = 0
    def write(self, b): self.n += len(b)
for it in range(2):
    sink = _Sink()
    t0 = time.perf_counter()
    al.align_file(fa, sink, batch_reads=n_reads // 4, paired=pe)
    dt = time.perf_counter() - t0
    print("%s: align_file, %d reads in 4 batches, FASTA on disk -> %d bytes of SAM: %.1f ms = %.2f Mreads/s" % ("PE" if pe else "SE", n_reads, sink.n, dt * 1e3, n_reads / dt / 1e6), flush=True)
    st = getattr(al, "last_stats", None)
    if st is not None:
        print("   native pipeline: %.1f ms for %d reads in %d batches on %d lanes = %.2f Mreads/s (reads in host memory -> text handed to the sink); writer: format %.1f ms; "
              "lanes (summed): H2D %.1f, seeding %.1f, chain+extend+merge %.1f, tail %.1f, select %.1f, CIGAR + D2H %.1f ms" %
              (st.seconds * 1e3, st.n_reads, st.n_batches, st.n_lanes, st.n_reads / st.seconds / 1e6, st.format_seconds * 1e3, st.h2d_seconds * 1e3, st.seed_seconds * 1e3,
               st.chain_extend_seconds * 1e3, st.tail_seconds * 1e3, st.select_seconds * 1e3, st.cigar_seconds * 1e3), flush=True)
