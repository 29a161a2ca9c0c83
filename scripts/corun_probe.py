#!/usr/bin/env python3
"""Do the gather-bound seeding of one batch and the VALU-bound extension of another share the chip productively?  The bench workload
(hg38-scale index, 1 M x 150 bp): seeding of batch B alone, chaining + extension of batch A alone, then both at once on two streams
(seeding on a high-priority stream or not); CORUN_CONFIGS sweeps the library's knobs (bmh_tune_set) inside one process.
usage: corun_probe.py [genome_mbp] [reads]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bwa-mem_gpu_amd"))
import numpy as np, torch
import bwamem_hip as B
from bwamem_hip import fmindex as F, pipeline as P
from bwamem_hip.lib import ChainWorkspace

mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 3100
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
dev = torch.device("cuda", 0)
L = B.load_library()
n_genome = int(mbp * 1e6)
g_t, meta = B.synth.make_genome_device(n_genome, dev, seed=42, return_meta=True)
pac_t = F.pack_pac_device(g_t); g = g_t.cpu().numpy(); del g_t
d = F.build_fmd_index_device(pac_t, n_genome, sa_intv=1, verify=False)
dindex = B.Index.from_device(d.primary, d.L2.astype(np.uint64), d.seq_len, d.bwt_t, d.sa_intv, d.sa_t, d.bits_t, pac_t=pac_t, l_pac=n_genome)
batches = [P.reads_to_device(B.synth.make_reads(g, n_reads, 150, seed=s, holes=meta["holes"])[0], dev) for s in (7, 1007)]
params = B.ExtParams.default()
wsA, wsB = B.SeedWorkspace(n_reads, n_reads * 150), B.SeedWorkspace(n_reads, n_reads * 150)
sA = wsA.seed_batch(dindex, batches[0].ascii, batches[0].offs, batches[0].lens, 19)
cw = ChainWorkspace(n_reads, int(sA.n_seeds * 1.25) + 4096)
cw.set_contigs(meta["contigs"]); cw.set_materialize(False)
dj = cw.chain_batch(dindex, batches[0].ascii, batches[0].offs, batches[0].lens, sA)
out = torch.zeros(int(dj.n_jobs) + 4096, 3, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
import threading


def timeit(fs, n=4):
    for f in fs: f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        for f in fs: f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3


# knob settings to sweep (bmh_tune_set; "-" = the library's defaults): CORUN_CONFIGS="EXT_PERSIST=3;EXT_PERSIST=2,SEED_LDS_PAD=20000;..."
configs = [c for c in os.environ.get("CORUN_CONFIGS", "-;EXT_PERSIST=3;EXT_PERSIST=2;EXT_PERSIST=1").split(";") if c]
prios = [int(x) for x in os.environ.get("CORUN_PRIOS", "0,-1").split(",")]
for cfg in configs:
    kv = [] if cfg == "-" else [x.split("=") for x in cfg.split(",")]
    for k, v in kv:
        L.bmh_tune_set(k.encode(), int(v), 0)
    for prio in prios:
        s_ext, s_seed = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev, priority=prio)
        ext = lambda: cw.extend(out, params=params, stream=s_ext.cuda_stream)
        seed = lambda: wsB.seed_batch(dindex, batches[1].ascii, batches[1].offs, batches[1].lens, 19, stream=s_seed.cuda_stream)
        te, ts = timeit([ext]), timeit([seed])

        # both at once: the seeding call returns to the host after its last internal synchronisation, so it is issued from a second thread
        def both():
            th = threading.Thread(target=lambda: (L.bmh_set_device(0), torch.cuda.set_device(0), seed()))
            th.start(); ext(); th.join()
        tb = timeit([both])
        if os.environ.get("CORUN_TRACE"):                  # one more co-run under the wave residency trace (csrc/wtrace.h)
            sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
            from wave_residency import trace, summary
            with trace(L) as tr:
                both(); torch.cuda.synchronize()
            print(summary(tr.records, title=f"[{cfg}] co-run: "), flush=True)
        print(f"{cfg:40s} seeding stream priority {prio:2d}: extension alone {te:6.2f} ms, seeding alone {ts:6.2f} ms, sum {te + ts:6.2f}, "
              f"both at once {tb:6.2f} ms ({(te + ts - tb) / min(te, ts) * 100:.0f} % of the shorter one hidden)", flush=True)
    for k, v in kv:
        L.bmh_tune_set(k.encode(), 0, 1)
