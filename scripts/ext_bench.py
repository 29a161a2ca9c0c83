#!/usr/bin/env python3
"""Extension-only microbenchmark: n synthetic jobs of one query length (diverged flank-like pairs), bmh_extend_batch timed with the
packed 16-bit kernels on and off, checksums compared.  Small enough in dispatch count to run under rocprofv3 --pmc.
usage: ext_bench.py [qlen=120] [n=1000000] [div=0.12] [h0=25] [reps=3] [modes=1,0]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bwa-mem_gpu_amd"))
import numpy as np, torch
import bwamem_hip as B

qlen = int(sys.argv[1]) if len(sys.argv) > 1 else 120
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
div = float(sys.argv[3]) if len(sys.argv) > 3 else 0.12
h0v = int(sys.argv[4]) if len(sys.argv) > 4 else 25
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
modes = [int(x) for x in (sys.argv[6] if len(sys.argv) > 6 else "1,0").split(",")]
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev); gen.manual_seed(5)
tlen = qlen + 60
t = torch.randint(0, 4, (n, tlen), dtype=torch.uint8, device=dev, generator=gen)
# query = target prefix, shifted by 0..2 behind a random position (a short deletion), substitutions at rate div
pos = torch.randint(0, qlen, (n, 1), device=dev, generator=gen)
sh = torch.randint(0, 3, (n, 1), device=dev, generator=gen) * (torch.rand(n, 1, device=dev, generator=gen) < 0.3)
j = torch.arange(qlen, device=dev)[None, :]
src = j + torch.where(j >= pos, sh, torch.zeros_like(sh))
q = torch.gather(t, 1, src)
mut = torch.rand(n, qlen, device=dev, generator=gen) < div
q = torch.where(mut, (q + torch.randint(1, 4, (n, qlen), dtype=torch.uint8, device=dev, generator=gen)) & 3, q).contiguous()
qoff = (torch.arange(n, device=dev, dtype=torch.int64) * qlen).to(torch.int32)
toff = (torch.arange(n, device=dev, dtype=torch.int64) * tlen).to(torch.int32)
ql = torch.full((n,), qlen, dtype=torch.int32, device=dev); tl = torch.full((n,), tlen, dtype=torch.int32, device=dev)
h0 = torch.full((n,), h0v, dtype=torch.int32, device=dev)
out = torch.zeros(n, 3, dtype=torch.int32, device=dev)
L = B.load_library()
# (BMH_EXT_STATS=1 in the environment prints rows executed, but its three atomics per job on one address cost ~33 ms per 1 M jobs: never time with it)
sums = {}
for mode in modes:
    L.bmh_extend_set_packed(mode)
    for it in range(reps):
        out.zero_()
        torch.cuda.synchronize(); t0 = time.time()
        B.extend_batch(q.view(-1), qoff, ql, t.view(-1), toff, tl, h0, out)
        torch.cuda.synchronize(); dt = time.time() - t0
        print(f"packed={mode} qlen {qlen} n {n}: {dt * 1e3:.2f} ms wall, {L.bmh_extend_last_ms():.2f} ms events", flush=True)
    sums[mode] = int(out.to(torch.int64).sum().item()), float(out[:, 0].float().mean().item())
print("checksums", sums, "identical" if len(set(sums.values())) == 1 else "DIFFERENT")
