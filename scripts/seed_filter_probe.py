"""Cost of the reference's seed filter (-W) in the device job builder: aligner stage times with and without it.
usage: seed_filter_probe.py [genome_mbp] [n_reads] [W]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bwa-mem_gpu_amd"))
import numpy as np, torch
import bwamem_hip as B
from bwamem_hip.aligner import Aligner, ReadSet
mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 200
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
W = sys.argv[3] if len(sys.argv) > 3 else "5"
g = B.synth.make_genome(int(mbp * 1e6), seed=42, repeat_frac=0.3)
idx = B.fmindex.build_fmd_index(g, device="cuda:0")
torch.cuda.empty_cache()
os.environ["BMH_ALIGNER_PROFILE"] = "1"
al = Aligner.from_memory(idx, g)
reads = B.synth.make_reads(g, n_reads, 150, seed=7, sub_rate=0.02)[0]
asc = B.synth.codes_to_ascii(reads)
rs = ReadSet.from_lists(["r%d" % i for i in range(n_reads)], [asc[i] for i in range(n_reads)])
for opts in ([], ["-W", W], [], ["-W", W]):
    al.set_options(opts if opts else ["-W", "0"])
    t0 = time.perf_counter()
    txt = al.align_batch(rs, id0=0, as_bytes="view")
    print("options %s: %d reads in %.1f ms" % (opts, n_reads, (time.perf_counter() - t0) * 1e3), flush=True)
