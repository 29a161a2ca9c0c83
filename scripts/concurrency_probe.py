import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bwa-mem_gpu_amd"))
import numpy as np, torch
import bwamem_hip as B
dev = torch.device("cuda:0")
gs = 200_000_000
nblk = (2 * gs + 63) // 64 + 1
bwt = torch.randint(-2**31, 2**31 - 1, (nblk * 8,), dtype=torch.int32, device=dev)
n_sa = (2 * gs + 16) // 16
sa = torch.zeros(n_sa, dtype=torch.int32, device=dev); bits = torch.zeros(n_sa // 32 + 1, dtype=torch.int32, device=dev)
L2 = np.array([0, gs // 2, gs, gs + gs // 2, 2 * gs], dtype=np.uint64)
idx = B.Index.from_device(12345, L2, 2 * gs, bwt, 16, sa, bits)
L = B.load_library()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
hip = C.CDLL("libamdhip64.so")
import threading
def run(stream, lanes, iters, out):
    ms = C.c_float(); L.bmh_calib_gather(idx.handle, lanes, iters, 1, C.c_void_p(stream.cuda_stream), C.byref(ms)); out.append(ms.value)
for lanes, iters in ((65536, 4096), (1 << 22, 64)):
    o = []; run(s1, lanes, iters, o); run(s1, lanes, iters, o)
    print("lanes %d iters %d alone: %.2f ms" % (lanes, iters, o[-1]))
    # two host threads, each launching on its own stream (the calib call blocks on its own event)
    o1, o2 = [], []
    t = time.time()
    th = [threading.Thread(target=run, args=(s1, lanes, iters, o1)), threading.Thread(target=run, args=(s2, lanes, iters, o2))]
    [x.start() for x in th]; [x.join() for x in th]
    print("  two streams concurrently: wall %.2f ms, kernel times %.2f / %.2f ms" % ((time.time() - t) * 1e3, o1[0], o2[0]))
