"""bmh_aligner_run over and over on the same reads (300 Mbp genome, 400 000 reads in four batches): device memory in use and host RSS must not grow run after run."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bwa-mem_gpu_amd"))
import numpy as np, torch
import bwamem_hip as B
from bwamem_hip import fmindex as F
from bwamem_hip.aligner import ReadSet
from bwamem_hip.lib import NativeAligner, PeOpt, ChainOpt, PostOpt
dev = torch.device("cuda:0"); L = B.load_library()
n_genome = 300_000_000
g_t, meta = B.synth.make_genome_device(n_genome, dev, seed=42, return_meta=True)
pac_t = F.pack_pac_device(g_t); g = g_t.cpu().numpy(); del g_t; torch.cuda.empty_cache()
d = F.build_fmd_index_device(pac_t, n_genome, sa_intv=1)
dindex = B.Index.from_device(d.primary, d.L2.astype(np.uint64), d.seq_len, d.bwt_t, d.sa_intv, d.sa_t, d.bits_t, pac_t=pac_t, l_pac=n_genome)
co = ChainOpt(); L.bmh_chain_opt_default(C.byref(co)); po = PostOpt(); L.bmh_post_opt_default(C.byref(po)); pe_o = PeOpt(); L.bmh_pe_opt_default(C.byref(pe_o))
nat = NativeAligner(dindex, pac_t.cpu().numpy(), n_genome, meta["contigs"], None, co, B.ExtParams.default(), po, pe_o)
n_reads, rl = 400_000, 150
import resource
for paired in (False, True):
    reads = (B.synth.make_pairs(g, n_reads // 2, rl, seed=3, holes=meta["holes"]) if paired else B.synth.make_reads(g, n_reads, rl, seed=3, holes=meta["holes"]))[0]
    flat = np.ascontiguousarray(np.asarray(reads, np.uint8).reshape(-1)); w = len(str(n_reads))
    names = np.char.add("r", np.char.zfill((np.arange(n_reads) // (2 if paired else 1)).astype(str), w))
    blob = np.frombuffer(("\0".join(names.tolist()) + "\0").encode(), dtype=np.uint8)
    rs = ReadSet(B.synth.codes_to_ascii(flat), np.arange(n_reads, dtype=np.uint64) * np.uint64(rl), np.full(n_reads, rl, np.uint32), blob, np.arange(n_reads, dtype=np.uint64) * np.uint64(w + 2), codes=flat)
    cuts = [0, 100_000, 200_000, 300_000, n_reads]
    for it in range(12):
        nat.run(rs, cuts, paired, lambda mv: None, n_lanes=3 if paired else 2, n_threads=16)
        if it in (1, 5, 11):
            free, tot = torch.cuda.mem_get_info()
            print("paired" if paired else "single", "run", it, "GPU memory in use %.1f MB" % ((tot - free) / 1e6), "host RSS %.0f MB" % (resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e3), flush=True)
nat.free()
free, tot = torch.cuda.mem_get_info(); print("after free: GPU memory in use %.1f MB" % ((tot - free) / 1e6))
