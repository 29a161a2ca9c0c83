"""Times the device job builder on the bench workload: seeding -> bmh_chain_batch -> bmh_extend_batch -> bmh_chain_merge."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bwa-mem_gpu_amd"))
import numpy as np, torch
import bwamem_hip as B
from bwamem_hip import pipeline as P
from bwamem_hip.lib import ChainWorkspace, load_library
mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 1000
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
L = int(sys.argv[3]) if len(sys.argv) > 3 else 150
dev = torch.device("cuda:0")
g = B.synth.make_genome(int(mbp * 1e6), seed=42)
idx = B.fmindex.build_fmd_index(g, device="cuda:0")
torch.cuda.empty_cache()
bwt, sa, bits = P.index_to_device_tensors(idx, dev)
pad = (-len(g)) % 4
codes = torch.from_numpy(np.concatenate([g, np.zeros(pad + 4, np.uint8)])).to(dev).view(-1, 4).to(torch.int32)
pac = ((codes[:, 0] << 6) | (codes[:, 1] << 4) | (codes[:, 2] << 2) | codes[:, 3]).to(torch.uint8).contiguous()
dindex = B.Index.from_device(idx.primary, idx.L2, idx.seq_len, bwt, idx.sa_intv, sa, bits, pac_t=pac, l_pac=len(g))
reads, _ = B.synth.make_reads(g, n_reads, L, seed=7)
dr = P.reads_to_device(reads, dev)
ws = B.SeedWorkspace(n_reads, n_reads * L)
s = ws.seed_batch(dindex, dr.ascii, dr.offs, dr.lens, 19)
cw = ChainWorkspace(n_reads, int(s.n_seeds) + 1024)
from bwamem_hip.lib import seeds_to_host
nr = seeds_to_host(s, n_reads)["n_ref_pos"]
os.environ["BMH_CHAIN_PROF_READ"] = str(int(nr.argmax()))
print("seeds/read: max", nr.max(), "top", np.sort(nr)[-8:], ">32:", int((nr > 32).sum()), ">64:", int((nr > 64).sum()), ">128:", int((nr > 128).sum()), ">256:", int((nr > 256).sum()))
lib = load_library()
SWEEP = os.environ.get("BMH_PROBE_SWEEP") is not None           # try several lane/wave thresholds of the device job builder
for it in range(12 if SWEEP else 4):
    os.environ["BMH_CHAIN_HEAVY"] = ["32", "16", "64", "128", "256", "100000"][it // 2] if SWEEP else "32"
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dj = cw.chain_batch(dindex, dr.ascii, dr.offs, dr.lens, s)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("heavy>%s chain_batch %.3f ms  jobs %d regs %d heavy %d qbytes %d tbytes %d" % (os.environ["BMH_CHAIN_HEAVY"], (t1 - t0) * 1e3, dj.n_jobs, dj.n_regs, dj.n_heavy_reads, dj.q_bytes, dj.t_bytes), flush=True)
out3 = torch.zeros(int(dj.n_jobs), 3, dtype=torch.int32, device=dev)
regs = torch.zeros(int(dj.n_regs), 8, dtype=torch.int32, device=dev)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lib.bmh_extend_batch(dj.d_q, dj.d_qoff, dj.d_qlen, dj.d_t, dj.d_toff, dj.d_tlen, dj.d_h0, int(dj.n_jobs), C.byref(B.ExtParams.default()), out3.data_ptr(), None, None)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    cw.merge(out3, regs)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("extend %.3f ms  merge %.3f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
r = regs.cpu().numpy()
print("regs checksum", int(r.astype(np.int64).sum()), "best score mean", float(np.maximum.reduceat(r[:, 1], np.unique(r[:, 0], return_index=True)[1]).mean()))
# CIGAR stage on the batch's regions: all of them, and one (the first) per read
from bwamem_hip.lib import cigar_batch
nr = int(dj.n_regs)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    cg, aln, md = cigar_batch(dindex, dr.ascii, dr.offs, dr.lens, regs, nr, max_cigar=16, md_cap=64)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("cigar_batch all %d regions: %.3f ms" % (nr, (t1 - t0) * 1e3), flush=True)
a = aln.cpu().numpy()
print("flags", np.bincount(a[:, 7], minlength=9)[:9], "mean n_cigar", a[:, 3].mean(), "mean NM", a[:, 4].mean())
rd = regs[:, 0].cpu().numpy()
first = np.unique(rd, return_index=True)[1].astype(np.int32)
sel = torch.from_numpy(first).to(dev)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    cg, aln, md = cigar_batch(dindex, dr.ascii, dr.offs, dr.lens, regs, len(first), sel_t=sel, max_cigar=16, md_cap=64)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("cigar_batch first region of %d reads: %.3f ms" % (len(first), (t1 - t0) * 1e3), flush=True)
