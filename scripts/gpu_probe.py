"""Scratch measurements on the GPU box: index build time, per-stage seeding time, extension time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bwa-mem_gpu_amd"))
import numpy as np, torch
import bwamem_hip as B
from bwamem_hip import pipeline as P

gsize = int(float(sys.argv[1])) if len(sys.argv) > 1 else 16_000_000
nreads = int(float(sys.argv[2])) if len(sys.argv) > 2 else 200_000
L = int(sys.argv[3]) if len(sys.argv) > 3 else 150
dev = torch.device("cuda:0")
t = time.time(); g = B.synth.make_genome(gsize, seed=42); print("genome %.1fs" % (time.time() - t), flush=True)
t = time.time(); idx = B.fmindex.build_fmd_index(g, device="cuda:0"); torch.cuda.synchronize(); print("index build %.1fs seq_len %d" % (time.time() - t, idx.seq_len), flush=True)
print("peak mem GB", torch.cuda.max_memory_allocated() / 1e9)
t = time.time(); reads, _ = B.synth.make_reads(g, nreads, L, seed=7); print("reads %.1fs" % (time.time() - t), flush=True)
bwt, sa, bits = P.index_to_device_tensors(idx, dev)
dindex = B.Index.from_device(idx.primary, idx.L2, idx.seq_len, bwt, idx.sa_intv, sa, bits)
dr = P.reads_to_device(reads, dev)
ws = B.SeedWorkspace(nreads, nreads * L)
for it in range(3):
    torch.cuda.synchronize(); t = time.time()
    s = ws.seed_batch(dindex, dr.ascii, dr.offs, dr.lens, 19)
    torch.cuda.synchronize(); dt = time.time() - t
    print("seed_batch %.1f ms wall; stages" % (dt * 1e3), {k: round(v, 2) for k, v in ws.timing().items()}, "seeds", s.n_seeds, "smems", s.n_smems, "cands", s.n_cands, flush=True)
seeds = P.seeds_to_torch(s, nreads, dev)
gf = torch.from_numpy(g).to(dev)
t = time.time(); jobs = P.first_seed_jobs(seeds, dr, gf); torch.cuda.synchronize(); print("job build %.2fs, %d jobs, qbases %d tbases %d" % (time.time() - t, jobs.n, jobs.q.numel(), jobs.t.numel()), flush=True)
print("qlen mean %.1f tlen mean %.1f" % (jobs.qlen.float().mean().item(), jobs.tlen.float().mean().item()))
out = torch.zeros(jobs.n, 3, dtype=torch.int32, device=dev)
for it in range(3):
    torch.cuda.synchronize(); t = time.time()
    B.extend_batch(jobs.q, jobs.qoff, jobs.qlen, jobs.t, jobs.toff, jobs.tlen, jobs.h0, out)
    torch.cuda.synchronize(); dt = time.time() - t
    print("extend %.1f ms for %d jobs -> %.2f Mjobs/s" % (dt * 1e3, jobs.n, jobs.n / dt / 1e6), flush=True)
cells = (jobs.qlen.long() * jobs.tlen.long()).sum().item()
print("upper-bound cells %.3g -> %.1f GCUPS" % (cells, cells / dt / 1e9))
print("to-end frac", (out[:, 1] == jobs.qlen).float().mean().item(), "mean score", out[:, 0].float().mean().item())
