"""Random 32-byte block gather ceiling + FETCH_SIZE calibration (run plain, and under rocprofv3 --pmc FETCH_SIZE)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bwa-mem_gpu_amd"))
import torch
import bwamem_hip as B
from bwamem_hip import pipeline as P
gs = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
dev = torch.device("cuda:0")
# a random "index" of the right size is enough: contents do not matter for the access pattern
nblk = (2 * gs + 63) // 64 + 1
bwt = torch.randint(-2**31, 2**31 - 1, (nblk * 8,), dtype=torch.int32, device=dev)
n_sa = (2 * gs + 16) // 16
sa = torch.zeros(n_sa, dtype=torch.int32, device=dev); bits = torch.zeros(n_sa // 32 + 1, dtype=torch.int32, device=dev)
import numpy as np
L2 = np.array([0, gs // 2, gs, gs + gs // 2, 2 * gs], dtype=np.uint64)
idx = B.Index.from_device(12345, L2, 2 * gs, bwt, 16, sa, bits)
L = B.load_library()
ms = C.c_float()
gs_note = "(index of %.1f GB)" % (nblk * 32 / 1e9)
# the width sweep (round 4): random requests of 32 / 64 / 128 aligned bytes -- is the ceiling in requests or in bytes?
for width in (1, 2, 4):
    lanes, iters = 1 << 22, 64
    for _ in range(2):
        L.bmh_calib_gather(idx.handle, lanes, iters, 1 | (width << 8), None, C.byref(ms))
    n = lanes * iters
    print(f"dependent gathers of {32 * width:3d} aligned bytes {gs_note}: {ms.value:.3f} ms, {n/ms.value/1e6:.1f} G requests/s, {n*32*width/ms.value/1e6:.0f} GB/s requested", flush=True)
# ... and of 16 bytes, ONE load instruction per request (round 5): is a request paid per load instruction (then a rank structure that needs one 16-byte
# load per rank would lift the seeding kernels' ceiling) or per cache line touched?
for dep in (1, 0):
    lanes, iters = 1 << 22, 64
    for _ in range(2):
        L.bmh_calib_gather(idx.handle, lanes, iters, dep | (1 << 16), None, C.byref(ms))
    n = lanes * iters
    print(f"{'dependent' if dep else 'independent'} gathers of  16 bytes, one load instruction {gs_note}: {ms.value:.3f} ms, {n/ms.value/1e6:.1f} G requests/s", flush=True)
for dep in (0, 1):
    for lanes, iters in ((1 << 20, 128), (1 << 22, 64), (1 << 23, 32)):
        L.bmh_calib_gather(idx.handle, lanes, iters, dep, None, C.byref(ms))  # warm
        L.bmh_calib_gather(idx.handle, lanes, iters, dep, None, C.byref(ms))
        n = lanes * iters
        print(f"dependent={dep} lanes={lanes} iters={iters}: {ms.value:.3f} ms, {n/ms.value/1e6:.1f} G gathers/s, {n*32/ms.value/1e6:.0f} GB/s (32B), {n*64/ms.value/1e6:.0f} GB/s (64B sectors)", flush=True)
