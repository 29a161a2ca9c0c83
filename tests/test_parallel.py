"""N > 1 path on CPU: world_size-2 gloo processes shard the reads, receive the index by broadcast,
and their per-shard outputs concatenate to the single-process result (checker = oracle)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import common

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_covers_everything():
    from bwamem_hip.parallel import shard_range
    for n in (0, 1, 7, 1000, 1001):
        for world in (1, 2, 3, 8):
            for mult in (1, 2):
                spans = [shard_range(n, r, world, mult) for r in range(world)]
                assert spans[0][0] == 0 and spans[-1][1] == n
                for (a, b), (c, d) in zip(spans, spans[1:]):
                    assert b == c and a <= b
                    assert b % mult == 0


def _worker(rank, world, port, tmp):
    sys.path.insert(0, os.path.join(ROOT, "bwa-mem_gpu_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_py
    from bwamem_hip import fmindex, synth
    from bwamem_hip.parallel import broadcast_index, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = synth.make_genome(60_000, seed=4)
    idx = fmindex.build_fmd_index(g) if rank == 0 else None
    hdr, bwt, sa, bits = broadcast_index(idx, torch.device("cpu"), src=0, world=world)
    # rebuild a host FMDIndex from what arrived and seed this rank's shard with the checker
    got = fmindex.FMDIndex(hdr["primary"], hdr["L2"].astype(np.int64), hdr["seq_len"], bwt.numpy().view(np.uint32),
                           hdr["sa_intv"], hdr["n_sa"], sa.numpy().view(np.uint32), bits.numpy().view(np.uint32), 1)
    reads, _ = synth.make_reads(g, 400, 100, seed=6)
    lo, hi = shard_range(reads.shape[0], rank, world)
    flat, offs, lens = common.flat_reads(reads[lo:hi])
    orc = oracle_py.Oracle()
    s = orc.seed_reads(orc.fmd(got), flat, offs, lens)
    np.savez(os.path.join(tmp, f"shard{rank}.npz"), **{k: s[k] for k in common.SEED_KEYS})
    dist.barrier()
    dist.destroy_process_group()


def _worker_built(rank, world, port, tmp):
    sys.path.insert(0, os.path.join(ROOT, "bwa-mem_gpu_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_py
    from bwamem_hip import fmindex, synth
    from bwamem_hip.parallel import broadcast_built_index, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cpu = torch.device("cpu")
    d = pac = meta = None
    if rank == 0:                          # what bench.py's rank 0 holds after bmh_index_build (here: the host builder's arrays)
        g_t, meta = synth.make_genome_device(60_000, cpu, seed=4, return_meta=True)
        d = fmindex.host_index_to_device_form(fmindex.build_fmd_index(g_t.numpy()), cpu)
        pac = fmindex.pack_pac_device(g_t)
    d, pac, meta = broadcast_built_index(d, pac, meta, cpu, src=0)
    g = fmindex.unpack_pac_device(pac, d.seq_len // 2).numpy()
    got = fmindex.device_index_to_host(d, d.sa_intv)
    reads, _ = synth.make_reads(g, 401, 100, seed=6, holes=meta["holes"])      # every rank draws the same reads, keeps its shard
    lo, hi = shard_range(reads.shape[0], rank, world)
    flat, offs, lens = common.flat_reads(reads[lo:hi])
    orc = oracle_py.Oracle()
    s = orc.seed_reads(orc.fmd(got), flat, offs, lens)
    np.savez(os.path.join(tmp, f"built{rank}.npz"), n_contigs=len(meta["contigs"]), **{k: s[k] for k in common.SEED_KEYS})
    dist.barrier()
    dist.destroy_process_group()


def _worker_own_text(rank, world, port, tmp):
    """bench.py's round-4 setup: every rank generates the (deterministic) genome itself and draws its reads while rank 0 builds the
    index; a checksum of the text is agreed on, then only the index arrays travel (broadcast_built_index(with_text=False))"""
    sys.path.insert(0, os.path.join(ROOT, "bwa-mem_gpu_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_py
    from bwamem_hip import fmindex, synth
    from bwamem_hip.parallel import broadcast_built_index, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cpu = torch.device("cpu")
    g_t, meta = synth.make_genome_device(60_000, cpu, seed=4, return_meta=True)           # on EVERY rank
    pac = fmindex.pack_pac_device(g_t)
    g = g_t.numpy()
    reads, _ = synth.make_reads(g, 401, 100, seed=6, holes=meta["holes"])                  # ... which can draw its reads before the index exists
    d = fmindex.host_index_to_device_form(fmindex.build_fmd_index(g), cpu) if rank == 0 else None
    chk = pac.sum(dtype=torch.int64).reshape(1) + (pac.numel() << 40)
    ref = chk.clone(); dist.broadcast(ref, 0)
    agree = (chk == ref).to(torch.int64); dist.all_reduce(agree, op=dist.ReduceOp.MIN)
    assert bool(agree.item())
    d, pac2, meta2 = broadcast_built_index(d, pac, meta, cpu, src=0, with_text=False)
    assert pac2 is pac and meta2 is meta
    got = fmindex.device_index_to_host(d, d.sa_intv)
    lo, hi = shard_range(reads.shape[0], rank, world)
    flat, offs, lens = common.flat_reads(reads[lo:hi])
    orc = oracle_py.Oracle()
    s = orc.seed_reads(orc.fmd(got), flat, offs, lens)
    np.savez(os.path.join(tmp, f"own{rank}.npz"), **{k: s[k] for k in common.SEED_KEYS})
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_index_only_broadcast_when_every_rank_has_the_text(tmp_path, oracle):
    from bwamem_hip import fmindex, synth
    from bwamem_hip.parallel import rebase_prefix
    port = 33500 + os.getpid() % 2000
    mp.spawn(_worker_own_text, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g_t, meta = synth.make_genome_device(60_000, torch.device("cpu"), seed=4, return_meta=True)
    g = g_t.numpy()
    idx = fmindex.build_fmd_index(g)
    reads, _ = synth.make_reads(g, 401, 100, seed=6, holes=meta["holes"])
    flat, offs, lens = common.flat_reads(reads)
    want = oracle.seed_reads(oracle.fmd(idx), flat, offs, lens)
    sh = [np.load(os.path.join(str(tmp_path), f"own{r}.npz")) for r in range(2)]
    for k in ("rbeg", "qbeg", "score", "n_ref_pos"):
        assert np.array_equal(np.concatenate([s[k] for s in sh]), want[k]), k
    assert np.array_equal(rebase_prefix([s["prefix"] for s in sh], [s["n_ref_pos"] for s in sh]), want["prefix"])


def test_two_rank_gloo_broadcast_of_the_device_built_index(tmp_path, oracle):
    """bench.py's N > 1 setup: rank 0 owns the index (blocks, suffix array, 2-bit pac, genome metadata) and broadcasts it;
    each rank rebuilds the genome from the pac it received, takes its read shard, and the shards concatenate to the
    single-process result"""
    from bwamem_hip import fmindex, synth
    from bwamem_hip.parallel import rebase_prefix
    port = 31500 + os.getpid() % 2000
    mp.spawn(_worker_built, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g_t, meta = synth.make_genome_device(60_000, torch.device("cpu"), seed=4, return_meta=True)
    g = g_t.numpy()
    idx = fmindex.build_fmd_index(g)
    reads, _ = synth.make_reads(g, 401, 100, seed=6, holes=meta["holes"])
    flat, offs, lens = common.flat_reads(reads)
    want = oracle.seed_reads(oracle.fmd(idx), flat, offs, lens)
    sh = [np.load(os.path.join(str(tmp_path), f"built{r}.npz")) for r in range(2)]
    assert all(int(s["n_contigs"]) == 24 for s in sh)
    for k in ("rbeg", "qbeg", "score", "n_ref_pos"):
        assert np.array_equal(np.concatenate([s[k] for s in sh]), want[k]), k
    assert np.array_equal(rebase_prefix([s["prefix"] for s in sh], [s["n_ref_pos"] for s in sh]), want["prefix"])


def test_two_rank_gloo_sharding_matches_single_process(tmp_path, oracle):
    from bwamem_hip import synth
    from bwamem_hip.parallel import rebase_prefix
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g, idx = common.genome_and_index(60_000, seed=4)
    reads, _ = synth.make_reads(g, 400, 100, seed=6)
    flat, offs, lens = common.flat_reads(reads)
    want = oracle.seed_reads(oracle.fmd(idx), flat, offs, lens)
    sh = [np.load(os.path.join(str(tmp_path), f"shard{r}.npz")) for r in range(2)]
    for k in ("rbeg", "qbeg", "score", "n_ref_pos"):
        assert np.array_equal(np.concatenate([s[k] for s in sh]), want[k]), k
    assert np.array_equal(rebase_prefix([s["prefix"] for s in sh], [s["n_ref_pos"] for s in sh]), want["prefix"])
