"""Probe: one 1 M-read batch as K sub-batches driven by K host threads on K streams (tails of one sub-batch's kernels filled by
the other's) vs the whole batch on one stream.  Same device stages as bench.py's default path."""
import os, sys, time, threading, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bwa-mem_gpu_amd"))
import bwamem_hip as B
from bwamem_hip import fmindex, synth
from bwamem_hip.lib import ChainWorkspace, SeedWorkspace, ExtParams
n_genome = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200_000_000
n_reads = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
dev = torch.device("cuda:0")
g = synth.make_genome(n_genome, seed=42)
idx = fmindex.build_fmd_index(g, device="cuda:0")
pad = (-len(g)) % 4
codes = np.concatenate([g, np.zeros(pad + 64, np.uint8)]).reshape(-1, 4)
pac = ((codes[:, 0] << 6) | (codes[:, 1] << 4) | (codes[:, 2] << 2) | codes[:, 3]).astype(np.uint8)
dindex = B.Index.upload(idx, pac=pac, l_pac=len(g)); dindex.densify_sa(1)
reads, _ = synth.make_reads(g, n_reads, 150, seed=7)
asc = synth.codes_to_ascii(reads.reshape(-1))
def run(K, reps=6):
    parts = []
    per = n_reads // K
    for k in range(K):
        lo, hi = k * per, (k + 1) * per if k < K - 1 else n_reads
        n = hi - lo
        r = torch.from_numpy(asc[lo * 150:hi * 150].copy()).to(dev)
        o = (torch.arange(n, dtype=torch.int64) * 150).to(torch.int32).to(dev); l = torch.full((n,), 150, dtype=torch.int32, device=dev)
        ws = SeedWorkspace(n, n * 150, max_cands=n * 60, max_occ=n * 16)
        st = torch.cuda.Stream(device=dev)
        parts.append(dict(n=n, r=r, o=o, l=l, ws=ws, st=st, cw=None, out3=None, regs=None))
    def work(p):
        sid = p["st"].cuda_stream
        s = p["ws"].seed_batch(dindex, p["r"], p["o"], p["l"], 19, stream=sid)
        if p["cw"] is None:
            p["cw"] = ChainWorkspace(p["n"], max(int(s.n_seeds) * 2, 1)); p["cw"].set_materialize(False)
        dj = p["cw"].chain_batch(dindex, p["r"], p["o"], p["l"], s, stream=sid)
        if p["out3"] is None:
            p["out3"] = torch.zeros(int(dj.n_jobs) * 2, 3, dtype=torch.int32, device=dev); p["regs"] = torch.zeros(int(dj.n_regs) * 2, 8, dtype=torch.int32, device=dev)
        p["cw"].extend(p["out3"], params=ExtParams.default(), stream=sid)
        p["cw"].merge(p["out3"], p["regs"], stream=sid)
        p["st"].synchronize()
    ts = []
    for rep in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter()
        th = [threading.Thread(target=work, args=(p,)) for p in parts]
        for x in th: x.start()
        for x in th: x.join()
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    print(f"K={K}: {1e3 * min(ts[2:]):.2f} ms best, {1e3 * np.median(ts[2:]):.2f} ms median", flush=True)
    for p in parts:
        p["cw"].free(); p["ws"].free()
for K in (1, 2, 3, 4, 1):
    run(K)
