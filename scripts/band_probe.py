"""Band geometry of the extension jobs of the bench workload (GPU box): what ksw_extend2's trimmed range [beg,end) looks like row
by row for every job the device job builder produces, so that column-window schemes can be costed on real data.

usage: python scripts/band_probe.py [genome_mbp=3100] [n_reads=20000] [read_len=150] [out=gpurun_out/band_probe.npz] [pe]
Writes the materialised job arrays (q, qoff, qlen, t, toff, tlen, h0, side, read) of the batch; scripts/band_model.py runs the
checker's per-row trace on them (CPU) and costs column-window / early-stop schemes.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bwa-mem_gpu_amd"))
import numpy as np
import torch
import bwamem_hip as B
from bwamem_hip import fmindex as F
from bwamem_hip import pipeline as P
from bwamem_hip.lib import ChainWorkspace, dev_jobs_to_host

mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 3100.0
n_reads = int(float(sys.argv[2])) if len(sys.argv) > 2 else 20000
rl = int(sys.argv[3]) if len(sys.argv) > 3 else 150
out = sys.argv[4] if len(sys.argv) > 4 else os.path.join(ROOT, "gpurun_out", "band_probe.npz")
paired = len(sys.argv) > 5 and sys.argv[5] == "pe"
dev = torch.device("cuda:0")
L = B.load_library()
n_genome = int(mbp * 1e6)
t0 = time.time()
g_t, meta = B.synth.make_genome_device(n_genome, dev, seed=42, return_meta=True)
pac_t = F.pack_pac_device(g_t)
del g_t
torch.cuda.empty_cache()
d = F.build_fmd_index_device(pac_t, n_genome, sa_intv=1, verify=False)
print(f"genome + index {time.time() - t0:.1f} s", flush=True)
dindex = B.Index.from_device(d.primary, d.L2.astype(np.uint64), d.seq_len, d.bwt_t, d.sa_intv, d.sa_t, d.bits_t, pac_t=pac_t, l_pac=n_genome)
g = F.unpack_pac_device(pac_t, n_genome).cpu().numpy()
reads, _ = (B.synth.make_pairs(g, n_reads // 2, rl, seed=1007, holes=meta["holes"]) if paired else B.synth.make_reads(g, n_reads, rl, seed=1007, holes=meta["holes"]))
dr = P.reads_to_device(reads, dev)
ws = B.SeedWorkspace(n_reads, n_reads * rl)
s = ws.seed_batch(dindex, dr.ascii, dr.offs, dr.lens, 19)
cw = ChainWorkspace(n_reads, int(s.n_seeds * 1.25) + 4096)
cw.set_contigs(meta["contigs"])
cw.set_materialize(True)
dj = cw.chain_batch(dindex, dr.ascii, dr.offs, dr.lens, s)
j = dev_jobs_to_host(dj, n_reads)
nj = len(j["qlen"])
print(f"{n_reads} reads: {int(s.n_seeds)} seeds, {nj} jobs, {int(dj.n_regs)} regions", flush=True)
arr = {k: np.ascontiguousarray(j[k]) for k in ("q", "qoff", "qlen", "t", "toff", "tlen", "h0", "job_side", "job_read")}
np.savez_compressed(out, **arr)
print("wrote", out, os.path.getsize(out))
