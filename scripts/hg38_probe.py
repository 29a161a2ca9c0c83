#!/usr/bin/env python3
"""The whole hot path on an hg38-scale index (seq_len > 2^32): hg38-like synthetic genome on the device -> bmh_index_build
(verified) -> reads -> seeding -> chaining/jobs -> extension -> regions; workload statistics (seeds, jobs per read), stage
times, parity against the oracle on the first reads and a truth check (the read's sampled position is among its regions).
usage: hg38_probe.py [mbp=3100] [n_reads=200000] [n_parity=2000]"""
import ctypes as C, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("bwa-mem_gpu_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import bwamem_hip as B
from bwamem_hip import fmindex as F, synth, pipeline as P
from bwamem_hip.lib import ChainWorkspace, HostJobs, seeds_to_host

mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 3100
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
n_par = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
dev = torch.device("cuda", 0)
L = B.load_library()
n = int(mbp * 1e6)
t0 = time.time()
g_t, meta = synth.make_genome_device(n, dev, seed=42, return_meta=True)
torch.cuda.synchronize(); print(f"genome {n} bp: {time.time()-t0:.1f}s planted {meta['planted']}", flush=True)
pac_t = F.pack_pac_device(g_t)
g = g_t.cpu().numpy(); del g_t; torch.cuda.empty_cache()
os.environ["BMH_BUILD_VERBOSE"] = "1"
t0 = time.time()
d = F.build_fmd_index_device(pac_t, n, sa_intv=1, verify=True)
print(f"index build {time.time()-t0:.1f}s {d.stats}", flush=True)
dindex = B.Index.from_device(d.primary, d.L2.astype(np.uint64), d.seq_len, d.bwt_t, 1, d.sa_t, d.bits_t, pac_t=pac_t, l_pac=n)
reads, truth = synth.make_reads(g, n_reads, 150, seed=7, holes=meta["holes"])
dr = P.reads_to_device(reads, dev)
ws = B.SeedWorkspace(n_reads, n_reads * 150)
for it in range(2):
    s = ws.seed_batch(dindex, dr.ascii, dr.offs, dr.lens, 19)
print("seeding", {k: round(v, 3) for k, v in ws.timing().items()}, f"seeds/read {s.n_seeds/n_reads:.1f} smems/read {s.n_smems/n_reads:.2f} cands/read {s.n_cands/n_reads:.1f}", flush=True)
cw = ChainWorkspace(n_reads, int(s.n_seeds * 1.25) + 4096)
cw.set_contigs(meta["contigs"])
cw.set_materialize(False)
torch.cuda.synchronize(); t0 = time.time()
dj = cw.chain_batch(dindex, dr.ascii, dr.offs, dr.lens, s)
torch.cuda.synchronize(); t_chain = time.time() - t0
nj, nr = int(dj.n_jobs), int(dj.n_regs)
out3 = torch.zeros(nj + 16, 3, dtype=torch.int32, device=dev); regs = torch.zeros(nr + 16, 8, dtype=torch.int32, device=dev)
cw.extend(out3); t_ext = L.bmh_extend_last_ms()
cw.merge(out3, regs); torch.cuda.synchronize()
print(f"chain {t_chain*1e3:.2f} ms, extend {t_ext:.2f} ms: jobs/read {nj/n_reads:.2f} regions/read {nr/n_reads:.2f} heavy reads {int(dj.n_heavy_reads)}", flush=True)
# truth: a region of the read covers the sampled position (either strand: text position p or 2n - p - len)
rg = regs[:nr].cpu().numpy()
rb = rg[:, 4].view(np.uint32).astype(np.int64) | (rg[:, 5].astype(np.int64) << 32)
re = rg[:, 6].view(np.uint32).astype(np.int64) | (rg[:, 7].astype(np.int64) << 32)
fb = np.where(rb >= n, 2 * n - re, rb); fe = np.where(rb >= n, 2 * n - rb, re)
tp = truth["pos"][rg[:, 0]]
hit = (fb < tp + 150) & (fe > tp)
found = np.zeros(n_reads, bool); found[rg[:, 0][hit]] = True
print(f"reads with a region over their true position: {found.mean()*100:.2f}%  (positions beyond 2^32 in regions: {(re >= 1 << 32).mean()*100:.1f}%)", flush=True)
# parity vs the oracle on the first reads
import oracle_py
orc = oracle_py.Oracle()
t0 = time.time()
hidx = F.device_index_to_host(d, 16)
print(f"host index copy {time.time()-t0:.1f}s", flush=True)
sub = reads[:n_par]; flat = np.ascontiguousarray(sub.reshape(-1)); offs = np.arange(n_par, dtype=np.uint64) * 150; lens = np.full(n_par, 150, np.uint32)
want = orc.seed_reads(orc.fmd(hidx), flat, offs, lens, 19, n_threads=8)
ws2 = B.SeedWorkspace(n_par, n_par * 150, max_cands=n_par * 150)
drs = P.reads_to_device(sub, dev)
s2 = ws2.seed_batch(dindex, drs.ascii, drs.offs, drs.lens, 19)
got = seeds_to_host(s2, n_par)
import common
common.assert_seeds_equal(got, want, "hg38-scale seeding: ")
print(f"seeding parity OK on {n_par} reads: {len(want['rbeg'])} seeds, max rbeg {int(want['rbeg'].max())} (2^32 = {1<<32})", flush=True)
cw2 = ChainWorkspace(n_par, int(s2.n_seeds) + 64); cw2.set_contigs(meta["contigs"]); cw2.set_materialize(False)
dj2 = cw2.chain_batch(dindex, drs.ascii, drs.offs, drs.lens, s2)
hj = HostJobs(g, flat, offs, lens, want, n_threads=8, contigs=meta["contigs"])
assert int(dj2.n_jobs) == hj.n_jobs and int(dj2.n_regs) == hj.n_regs, (int(dj2.n_jobs), hj.n_jobs, int(dj2.n_regs), hj.n_regs)
o3 = torch.zeros(hj.n_jobs + 1, 3, dtype=torch.int32, device=dev); r8 = torch.zeros(hj.n_regs + 1, 8, dtype=torch.int32, device=dev)
cw2.extend(o3); cw2.merge(o3, r8); torch.cuda.synchronize()
want3, _, _ = orc.extend_batch(*hj.jobs(), n_threads=8)
assert np.array_equal(o3.cpu().numpy()[: hj.n_jobs], want3), "extension results differ"
assert np.array_equal(r8.cpu().numpy()[: hj.n_regs], hj.merge(want3)), "regions differ"
print(f"chain/extend/merge parity OK: {hj.n_jobs} jobs, {hj.n_regs} regions", flush=True)
