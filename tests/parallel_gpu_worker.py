"""One rank of the N > 1 hot path on real hardware (started by tests/test_parallel_gpu.py as a fresh process per rank).

usage: parallel_gpu_worker.py <outdir> <genome_bp> <n_reads> <read_len> <paired 0|1>     (RANK / WORLD_SIZE / MASTER_* from the env)
What bench.py's ranks do, at test size: rank 0 builds genome and index in HBM, `broadcast_built_index` hands them to every rank
(over gloo through host memory when the ranks share one device, over RCCL when each has its own), every rank takes its
`shard_range` of the reads and runs the HIP path (bmh_seed_batch -> bmh_chain_extend_merge) on it, and writes its seeds and
regions to <outdir>/rank<r>.npz for the parent to concatenate.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bwa-mem_gpu_amd"))
import numpy as np
import torch
import torch.distributed as dist

import bwamem_hip as B
from bwamem_hip import fmindex as F, pipeline as P, synth
from bwamem_hip.lib import ChainWorkspace, seeds_to_host
from bwamem_hip.parallel import broadcast_built_index, shard_range


def main():
    outdir, n_genome, n_reads, rl, paired = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    n_dev = torch.cuda.device_count()
    dev_id = rank % n_dev
    torch.cuda.set_device(dev_id)
    dev = torch.device("cuda", dev_id)
    L = B.load_library()
    L.bmh_set_device(dev_id)
    backend = "gloo" if n_dev < world else "nccl"
    dist.init_process_group(backend, rank=rank, world_size=world, **({"device_id": dev} if backend == "nccl" else {}))
    d = pac_t = meta = None
    if rank == 0:
        g_t, meta = synth.make_genome_device(n_genome, dev, seed=21, return_meta=True)
        pac_t = F.pack_pac_device(g_t)
        del g_t
        d = F.build_fmd_index_device(pac_t, n_genome, sa_intv=1, verify=True)
    if backend == "gloo":                    # ranks share the device: through host memory, into each rank's own HBM allocation
        cpu = torch.device("cpu")
        if rank == 0:
            dc = F.DeviceFMDIndex(d.primary, d.L2, d.seq_len, d.bwt_t.cpu(), d.sa_intv, d.sa_t.cpu(), d.bits_t.cpu(), d.stats)
            broadcast_built_index(dc, pac_t.cpu(), meta, cpu, src=0)
        else:
            dc, pc, meta = broadcast_built_index(None, None, None, cpu, src=0)
            d = F.DeviceFMDIndex(dc.primary, dc.L2, dc.seq_len, dc.bwt_t.to(dev), dc.sa_intv, dc.sa_t.to(dev), dc.bits_t.to(dev), {})
            pac_t = pc.to(dev)
    else:
        d, pac_t, meta = broadcast_built_index(d, pac_t, meta, dev, src=0)
    dindex = B.Index.from_device(d.primary, d.L2.astype(np.uint64), d.seq_len, d.bwt_t, d.sa_intv, d.sa_t, d.bits_t, pac_t=pac_t, l_pac=n_genome)
    g = F.unpack_pac_device(pac_t, n_genome).cpu().numpy()
    if paired:
        reads, _ = synth.make_pairs(g, n_reads // 2, rl, seed=5, holes=meta["holes"])
    else:
        reads, _ = synth.make_reads(g, n_reads, rl, seed=5, holes=meta["holes"])      # every rank draws the same reads, keeps its shard
    lo, hi = shard_range(reads.shape[0], rank, world, multiple=2 if paired else 1)
    mine = np.ascontiguousarray(reads[lo:hi])
    n = hi - lo
    dr = P.reads_to_device(mine, dev)
    ws = B.SeedWorkspace(n, n * rl)
    s = ws.seed_batch(dindex, dr.ascii, dr.offs, dr.lens, 19)
    cw = ChainWorkspace(n, int(s.n_seeds) + 64)
    cw.set_contigs(meta["contigs"])
    cw.set_materialize(False)
    dj = cw.chain_batch(dindex, dr.ascii, dr.offs, dr.lens, s)          # sizes the output
    regs = torch.zeros(int(dj.n_regs) + 1, 8, dtype=torch.int32, device=dev)
    dj = cw.extend_merge(dindex, dr.ascii, dr.offs, dr.lens, s, regs)
    torch.cuda.synchronize()
    sh = seeds_to_host(s, n)
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), lo=lo, hi=hi, regs=regs.cpu().numpy()[: int(dj.n_regs)], n_jobs=int(dj.n_jobs), **sh)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
