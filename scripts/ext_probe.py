"""Extension-only probe on the bench workload: job-size statistics, kernel time, rows executed."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bwa-mem_gpu_amd"))
import numpy as np, torch
import bwamem_hip as B
from bwamem_hip import pipeline as P
from bwamem_hip.lib import HostJobs, seeds_to_host
gsize = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200_000_000
nreads = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
L = int(sys.argv[3]) if len(sys.argv) > 3 else 150
dev = torch.device("cuda:0")
g = B.synth.make_genome(gsize, seed=42)
idx = B.fmindex.build_fmd_index(g, device="cuda:0")
reads, _ = B.synth.make_reads(g, nreads, L, seed=7)
bwt, sa, bits = P.index_to_device_tensors(idx, dev)
dindex = B.Index.from_device(idx.primary, idx.L2, idx.seq_len, bwt, idx.sa_intv, sa, bits)
dr = P.reads_to_device(reads, dev)
ws = B.SeedWorkspace(nreads, nreads * L)
s = ws.seed_batch(dindex, dr.ascii, dr.offs, dr.lens, 19)
hj = HostJobs(g, reads.reshape(-1), np.arange(nreads, dtype=np.uint64) * L, np.full(nreads, L, np.uint32), seeds_to_host(s, nreads))
print("jobs %d regs %d" % (hj.n_jobs, hj.n_regs))
ql, tl = hj.qlen.astype(np.int64), hj.tlen.astype(np.int64)
print("qlen mean %.1f p50 %d p90 %d max %d | tlen mean %.1f p50 %d p90 %d max %d | cells(q*t) %.3g" % (ql.mean(), np.median(ql), np.percentile(ql, 90), ql.max(), tl.mean(), np.median(tl), np.percentile(tl, 90), tl.max(), float((ql * tl).sum())))
print("class histogram (ceil(qlen/16)):", np.bincount((ql + 15) // 16)[:12])
arrs = [torch.from_numpy(np.ascontiguousarray(x).view(np.int32) if x.dtype == np.uint32 else np.ascontiguousarray(x)).to(dev) for x in hj.jobs()]
out = torch.zeros(hj.n_jobs, 3, dtype=torch.int32, device=dev)
L_ = B.load_library()
for it in range(3):
    torch.cuda.synchronize(); t = time.time()
    B.extend_batch(*arrs, out)
    torch.cuda.synchronize(); dt = time.time() - t
    print("extend %.2f ms wall, %.2f ms events" % (dt * 1e3, L_.bmh_extend_last_ms()), flush=True)
