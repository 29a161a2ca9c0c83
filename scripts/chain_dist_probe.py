#!/usr/bin/env python3
"""Distribution of seeds per read on the hg38-like workload (what the chaining kernels face) + chain stage timing per heavy threshold.
usage: chain_dist_probe.py [mbp=3100] [n_reads=200000]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("bwa-mem_gpu_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import bwamem_hip as B
from bwamem_hip import fmindex as F, synth, pipeline as P
from bwamem_hip.lib import ChainWorkspace, seeds_to_host
mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 3100
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
dev = torch.device("cuda", 0)
L = B.load_library()
n = int(mbp * 1e6)
g_t, meta = synth.make_genome_device(n, dev, seed=42, return_meta=True)
pac_t = F.pack_pac_device(g_t)
g = g_t.cpu().numpy(); del g_t; torch.cuda.empty_cache()
d = F.build_fmd_index_device(pac_t, n, sa_intv=1)
dindex = B.Index.from_device(d.primary, d.L2.astype(np.uint64), d.seq_len, d.bwt_t, 1, d.sa_t, d.bits_t, pac_t=pac_t, l_pac=n)
reads, truth = synth.make_reads(g, n_reads, 150, seed=7, holes=meta["holes"])
dr = P.reads_to_device(reads, dev)
ws = B.SeedWorkspace(n_reads, n_reads * 150)
s = ws.seed_batch(dindex, dr.ascii, dr.offs, dr.lens, 19)
h = seeds_to_host(s, n_reads)
nref = h["n_ref_pos"].astype(np.int64)
sc = h["score"].astype(np.int64)
heads = np.nonzero(sc)[0]
read_of = np.repeat(np.arange(n_reads), nref)
need = np.zeros(n_reads, np.int64)
np.add.at(need, read_of[heads], np.minimum(sc[heads], 500))
qs = [50, 90, 95, 98, 99, 99.5, 99.9, 99.99, 100]
print("n_ref percentiles", {q: int(np.percentile(nref, q)) for q in qs})
print("need  percentiles", {q: int(np.percentile(need, q)) for q in qs})
for lo, hi in ((0, 8), (8, 32), (32, 64), (64, 128), (128, 256), (256, 512), (512, 1024), (1024, 4096), (4096, 1 << 40)):
    m = (need > lo) & (need <= hi)
    print(f"need in ({lo},{hi}]: {m.sum()} reads ({m.mean()*100:.2f}%), seeds {need[m].sum()} ({need[m].sum()/max(need.sum(),1)*100:.1f}% of sampled), n_ref sum {nref[m].sum()}")
cw = ChainWorkspace(n_reads, int(s.n_seeds * 1.25) + 4096)
cw.set_contigs(meta["contigs"]); cw.set_materialize(False)
for ht in (os.environ.get("BMH_CHAIN_HEAVY", "32").split(",")):
    os.environ["BMH_CHAIN_HEAVY"] = ht
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        dj = cw.chain_batch(dindex, dr.ascii, dr.offs, dr.lens, s)
        torch.cuda.synchronize(); t = time.time() - t0
    tm = (C_float4 := None)
    print(f"heavy>{ht}: chain {t*1e3:.2f} ms heavy reads {int(dj.n_heavy_reads)} jobs {int(dj.n_jobs)}", flush=True)
    if hasattr(L, "bmh_chain_last_timing"):
        import ctypes as C
        ms = (C.c_float * 8)(); L.bmh_chain_last_timing.argtypes = [C.c_void_p, C.POINTER(C.c_float)]; L.bmh_chain_last_timing(cw.handle, ms)
        print("   kernel ms:", [round(x, 3) for x in ms])
if os.environ.get("CHAIN_PROF"):
    # phase stamps of the largest reads of each class (needs a -DCH_PROFILE build: BMH_LIB=build/variants/lib_prof.so)
    for lo, hi in ((12, 16), (20, 32), (40, 64), (100, 128), (200, 256), (500, 512), (520, 620), (600, 1250), (1250, 1900)):
        cand = np.nonzero((need > lo) & (need <= hi))[0]
        if len(cand) == 0: continue
        r = int(cand[len(cand) // 2])
        os.environ["BMH_CHAIN_PROF_READ"] = str(r)
        print(f"read {r}: need {need[r]} n_ref {nref[r]}", flush=True)
        cw.chain_batch(dindex, dr.ascii, dr.offs, dr.lens, s)
