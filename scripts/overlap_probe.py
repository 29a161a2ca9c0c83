"""Do a gather-bound kernel and the VALU-bound extension kernels overlap when issued on two streams?"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bwa-mem_gpu_amd"))
import numpy as np, torch
import bwamem_hip as B
from bwamem_hip import pipeline as P
from bwamem_hip.lib import HostJobs, seeds_to_host
dev = torch.device("cuda:0")
gsize, nreads, L = 200_000_000, 1_000_000, 150
g = B.synth.make_genome(gsize, seed=42)
idx = B.fmindex.build_fmd_index(g, device="cuda:0")
reads, _ = B.synth.make_reads(g, nreads, L, seed=7)
bwt, sa, bits = P.index_to_device_tensors(idx, dev)
dindex = B.Index.from_device(idx.primary, idx.L2, idx.seq_len, bwt, idx.sa_intv, sa, bits)
dr = P.reads_to_device(reads, dev)
ws = B.SeedWorkspace(nreads, nreads * L)
s = ws.seed_batch(dindex, dr.ascii, dr.offs, dr.lens, 19)
hj = HostJobs(g, reads.reshape(-1), np.arange(nreads, dtype=np.uint64) * L, np.full(nreads, L, np.uint32), seeds_to_host(s, nreads))
arrs = [torch.from_numpy(np.ascontiguousarray(x).view(np.int32) if x.dtype == np.uint32 else np.ascontiguousarray(x)).to(dev) for x in hj.jobs()]
out = torch.zeros(hj.n_jobs, 3, dtype=torch.int32, device=dev)
Lb = B.load_library()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
ms = C.c_float()
def ext(): B.extend_batch(*arrs, out, stream=s1.cuda_stream)
def seed(): ws.seed_batch(dindex, dr.ascii, dr.offs, dr.lens, 19, stream=s2.cuda_stream)
def timeit(f, n=5):
    f(); torch.cuda.synchronize(); t = time.time()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.time() - t) / n * 1e3
print("extension alone %.2f ms" % timeit(ext))
print("seeding alone   %.2f ms" % timeit(seed))
print("ext then seed (2 streams) %.2f ms" % timeit(lambda: (ext(), seed())))
# pure gather kernel on a raw HIP stream next to the extension
hip = C.CDLL("libamdhip64.so")
def gather(): Lb.bmh_calib_gather(dindex.handle, 1 << 22, 64, 1, C.c_void_p(s2.cuda_stream), C.byref(ms))
print("gather alone    %.2f ms" % timeit(gather))
print("ext + gather    %.2f ms" % timeit(lambda: (ext(), gather())))
