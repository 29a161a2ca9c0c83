"""Inputs of the paired-end region tail (bmh_finalize_pairs) of one batch, written to gpurun_out/pe_tail.npz for profiling the host code
off the GPU box (scripts/pe_tail_prof.cpp).  usage: dump_pe_tail.py [genome_mbp] [n_reads]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bwa-mem_gpu_amd"))
import numpy as np, torch
import bwamem_hip as B
from bwamem_hip import fmindex as F, pipeline as P
from bwamem_hip.lib import ChainWorkspace, dev_jobs_to_host
mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 100
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
dev = torch.device("cuda:0")
n_genome = int(mbp * 1e6)
g_t, meta = B.synth.make_genome_device(n_genome, dev, seed=42, return_meta=True)
pac_t = F.pack_pac_device(g_t)
g = g_t.cpu().numpy(); del g_t
d = F.build_fmd_index_device(pac_t, n_genome, sa_intv=1)
contigs, holes = meta["contigs"], meta["holes"]
dindex = B.Index.from_device(d.primary, d.L2.astype(np.uint64), d.seq_len, d.bwt_t, d.sa_intv, d.sa_t, d.bits_t, pac_t=pac_t, l_pac=n_genome)
rl = 150
reads = B.synth.make_pairs(g, n_reads // 2, rl, seed=7, holes=holes)[0]
dr = P.reads_to_device(reads, dev)
ws = B.SeedWorkspace(n_reads, n_reads * rl)
s = ws.seed_batch(dindex, dr.ascii, dr.offs, dr.lens, 19)
cw = ChainWorkspace(n_reads, int(s.n_seeds) + 4096); cw.set_contigs(contigs); cw.set_materialize(False)
dj = cw.chain_batch(dindex, dr.ascii, dr.offs, dr.lens, s)
out3 = torch.zeros(max(int(dj.n_jobs), 1), 3, dtype=torch.int32, device=dev); regs = torch.zeros(max(int(dj.n_regs), 1), 8, dtype=torch.int32, device=dev)
cw.extend(out3); cw.merge(out3, regs); torch.cuda.synchronize()
dh = dev_jobs_to_host(dj, n_reads)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "pe_tail.npz"), pac=pac_t.cpu().numpy(), l_pac=n_genome, reads=np.ascontiguousarray(reads.reshape(-1)), rl=rl, n_reads=n_reads,
                    regs=regs.cpu().numpy()[: int(dj.n_regs)], rpr=np.ascontiguousarray(dh["regs_per_read"]), fr=np.ascontiguousarray(dh["frac_rep"], dtype=np.float32),
                    ctg_len=np.asarray([c[1] for c in contigs], np.int32))
print("regions", int(dj.n_regs), "reads", n_reads)
