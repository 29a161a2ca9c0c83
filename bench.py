#!/usr/bin/env python3
"""bench.py -- seed-and-extend hot path on MI355X: Mreads/s, roofline and CPU baseline.

Contract (driver): python bench.py --gpus N --steps K --warmup W ; for N > 1 launched by
torch.distributed.run, one rank per GPU.  Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[1], the configuration the metric is quoted on):
  1 M synthetic 150 bp single-end reads per GPU against an hg38-SCALE index: a seeded synthetic 3.1 Gbp genome
  (24 contigs in GRCh38's proportions, 50 % planted repeats -- mid-copy families, a LINE-like and a high-copy SINE-like
  family, tandem satellites, low-divergence segmental duplications --, N-runs; bwamem_hip/synth.py make_genome_device),
  seq_len = 6.2e9 > 2^32 rows, built ON THE DEVICE by bmh_index_build in the run's setup (there is no hg38 on the box and
  no network) and verified completely (every adjacent pair of suffix-array rows compared).
  A step = one pass of the hot path over one batch, reads in -> alignment regions out, entirely on the device:
  SMEM seeding (pack, forward, backward, filter, expand, locate) -> chaining, chain filter and extension-job construction
  with on-device reference fetch (bmh_chain_batch) -> seed extension (ksw_extend2 kernels) -> region merge.  Steps take
  `--distinct-batches` (4) different read batches in turn; `--inflight` (2) batches are in flight at a time, each on its own stream with its own
  workspaces and host thread, the way the reference keeps several gpu_storage batches in flight across its host threads
  (src/fastmap.c:417-534); every timed batch goes through the whole path (--inflight 1: strictly one batch at a time).  `value` has the reads resident in HBM when the timed region starts (bench contract);
  `incl_pcie` times the same steps fed from pinned host memory (reads H2D, regions D2H, double-buffered on copy streams) --
  the reference's boundary, seed_gen.cu:1841-1843,2073-2101 -- and is reported beside it.
Multi-GPU: reads shard across ranks (weak scaling: READS_PER_GPU per rank); rank 0 builds the index and broadcasts it
over RCCL once, outside the timed region; no data-path collective.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "bwa-mem_gpu_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bwamem_hip as B  # noqa: E402
from bwamem_hip import fmindex as F  # noqa: E402
from bwamem_hip import pipeline as P  # noqa: E402
from bwamem_hip.parallel import broadcast_built_index, shard_range  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_LANEOPS = 256 * 4 * 32 * 2.4e9   # 256 CUs x 4 SIMD-32 x 2.4 GHz: a wave64 instruction issues in 2 cycles (MI355X_MICROARCH.md)
CACHE_VERSION = "v3"          # bump when synth.make_genome_device or the index layout changes
PROFILE_TAG = "r06"          # profiles/<tag>_pmc*.json: counters collected by scripts/profile_round.sh with this same command (one file per workload)


def profile_counters(workload_key: str, lib_sha: str):
    """Per-kernel-family counters of the committed rocprofv3 PMC passes (profiles/<tag>_pmc.json, written by
    scripts/summarize_profiles.py from separate --pmc runs of this command).  They are NOT measured by this run: every
    field taken from them is labelled from_profile and dropped when the profile's workload differs from this run's or when it was
    collected on another build of the library (lib_sources_sha16: hash of csrc/ + include/, bwamem_hip.lib.sources_sha16)."""
    import glob
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", PROFILE_TAG + "*_pmc.json"))):
        try:
            d = json.load(open(fn))
        except Exception:
            continue
        if d.get("workload_key") == workload_key and d.get("lib_sources_sha16") == lib_sha:
            d["_file"] = os.path.relpath(fn, ROOT)
            return d
    return None


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def effective_cores():
    """Host cores this process may actually use: the affinity mask, capped by the cgroup CPU quota (a GPU box hands a
    job a share of its host, e.g. 16 of 256 hardware threads; os.cpu_count() reports the whole machine)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(-(-int(txt[0]) // int(txt[1])))))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, -(-q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()))))
            break
        except Exception:
            continue
    return n


def cpu_baseline(g, pac_h, hidx, reads, contigs, n_all: int, n_one: int, n_threads: int):
    """The CPU path timed on the host cores on bounded samples of the same batch: our C restatement (oracle/, parity-pinned to
    the compiled reference) on all cores and on one thread, and -- where oracle/_ref/libref.so travelled -- the reference's own
    compiled bwt_smem1 + bwt_sa (src/bwt.c:483-566,105-115) and ksw_extend2 (src/ksw.c:864) on one thread on the same sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_py
    from bwamem_hip.lib import HostJobs
    orc = oracle_py.Oracle()
    f = orc.fmd(hidx)
    L = reads.shape[1]
    out = {}

    def run(n, nth, tag):
        sub = reads[:n]
        flat = np.ascontiguousarray(sub.reshape(-1)); offs = np.arange(n, dtype=np.uint64) * L; lens = np.full(n, L, np.uint32)
        t0 = time.time(); s = orc.seed_reads(f, flat, offs, lens, 19, n_threads=nth); t_seed = time.time() - t0
        t0 = time.time(); hj = HostJobs(g, flat, offs, lens, s, n_threads=nth, contigs=contigs, pac=pac_h); t_chain = time.time() - t0 - hj.t_pack
        arr = [x.copy() for x in hj.jobs()]; n_jobs = hj.n_jobs
        t0 = time.time(); o3, _, cells = orc.extend_batch(*arr, n_threads=nth); t_ext = time.time() - t0
        regs = hj.merge(o3).copy(); hj.free()             # (src/bwamem.c:2297-2303: the regions the reference's host code would hold after its extension)
        out[tag] = dict(n=n, threads=nth, t_seed=t_seed, t_chain=t_chain, t_ext=t_ext, n_jobs=n_jobs, cells=cells, n_seeds=int(len(s["rbeg"])), work=s["work"])
        if tag == "all":
            out["_check"] = dict(n=n, seeds=s, regs=regs)
        return flat, offs, lens, s, arr, o3

    run(n_all, n_threads, "all")
    if n_one <= 0:
        return out
    flat, offs, lens, s, arr, o3 = run(n_one, 1, "one")
    if oracle_py.Ref.available():
        ref = oracle_py.Ref()
        if hasattr(ref.lib, "ref_bwt_from_gpu_layout"):
            b = ref.bwt_from_index_fast(hidx)
            t0 = time.time(); rs = ref.seed_reads(b, flat, offs, lens, 19); t_seed = time.time() - t0
            same = all(np.array_equal(rs[k], s[k]) for k in ("rbeg", "qbeg", "score", "n_ref_pos"))
            t0 = time.time(); ro3, _ = ref.extend_batch(*arr); t_ext = time.time() - t0
            same = same and np.array_equal(ro3, o3)
            out["ref"] = dict(n=n_one, threads=1, t_seed=t_seed, t_ext=t_ext, identical_to_port=bool(same))
            ref.lib.ref_bwt_free(b)
    return out


def verify_against_oracle(check, gpu_seeds, gpu_regs):
    """The GPU's seeds and alignment regions of the first check['n'] reads of a batch against the CPU checker's (oracle seeding ->
    bmh_build_jobs -> oracle ksw_extend2 -> bmh_merge_regs, i.e. what the reference's host code holds at src/bwamem.c:2297-2303).
    Both sides are in read order, so the checker's arrays must equal the leading rows of the GPU's.  Returns the `verified` object."""
    n, s, regs = check["n"], check["seeds"], check["regs"]
    ns = int(len(s["rbeg"]))
    seeds_ok = bool(np.array_equal(gpu_seeds["n_ref_pos"][:n], s["n_ref_pos"]) and np.array_equal(gpu_seeds["prefix"][:n], s["prefix"]) and
                    np.array_equal(gpu_seeds["rbeg"][:ns], s["rbeg"]) and np.array_equal(gpu_seeds["qbeg"][:ns], s["qbeg"]) and
                    np.array_equal(gpu_seeds["score"][:ns], s["score"]))
    m = int(len(regs))
    regs_ok = bool(len(gpu_regs) >= m and np.array_equal(gpu_regs[:m], regs) and (len(gpu_regs) == m or int(gpu_regs[m, 0]) >= n))
    return {"reads": int(n), "seeds": ns, "regions": m, "seeds_identical": seeds_ok, "regions_identical": regs_ok}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--genome-mbp", type=float, default=float(os.environ.get("BENCH_GENOME_MBP", "3100")))
    ap.add_argument("--reads-per-gpu", type=int, default=int(os.environ.get("BENCH_READS_PER_GPU", "1000000")))
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--sa-intv", type=int, default=1, help="suffix-array samples resident in HBM: every N-th row (1 = the whole suffix array: "
                    "locating a seed is one gather; 16 = what the reference's `bwa index` writes for its GPU index, src/bwtindex.c:324)")
    ap.add_argument("--paired", action="store_true", help="interleaved 2 x read-len pairs (configs[3]); reads-per-gpu counts reads, shards stay on pair boundaries")
    ap.add_argument("--no-verify-index", dest="verify", action="store_false", help="skip the complete check of the suffix array after the build")
    ap.add_argument("--index-cache", default=os.environ.get("BENCH_INDEX_CACHE", ""), help="directory: save the built genome + index there / load them from there "
                    "(setup only; used by scripts/profile_round.sh so that the rocprofv3 passes do not rebuild it)")
    ap.add_argument("--inflight", type=int, default=int(os.environ.get("BENCH_INFLIGHT", "2")), help="batches in flight: each goes through the whole path on its own "
                    "stream, workspaces and host thread (1 = strictly one batch at a time)")
    ap.add_argument("--passes", type=int, default=int(os.environ.get("BENCH_PASSES", "2")), help="2: bmh_chain_extend_merge (the jobs of the lightly seeded reads are "
                    "extended while the seed-rich reads are still being chained, then theirs); 1: bmh_chain_batch -> bmh_chain_extend -> bmh_chain_merge "
                    "(one extension over all jobs of the batch, after all of its chaining: pays with batches in flight, whose work fills the wait)")
    ap.add_argument("--no-pcie", dest="pcie", action="store_false", help="skip the second timed loop (steps fed from pinned host memory)")
    ap.add_argument("--no-next-rows", dest="next_rows", action="store_false")
    ap.add_argument("--cpu-sample", type=int, default=int(os.environ.get("BENCH_CPU_SAMPLE", "100000")), help="reads of the all-cores CPU leg (0 = no CPU baseline)")
    ap.add_argument("--cpu-sample-1t", type=int, default=int(os.environ.get("BENCH_CPU_SAMPLE_1T", "3000")), help="reads of the one-thread CPU legs")
    ap.add_argument("--verify-sample", type=int, default=int(os.environ.get("BENCH_VERIFY_SAMPLE", "-1")), help="reads of the last timed batch whose GPU seeds and regions are "
                    "compared with the oracle after the timed loops (-1: the CPU sample at N = 1, 10000 per rank at N > 1; 0: off)")
    ap.add_argument("--print-cache-dir", action="store_true", help="print the --index-cache directory this command would use and exit (for the profile scripts)")
    ap.add_argument("--distinct-batches", type=int, default=int(os.environ.get("BENCH_DISTINCT_BATCHES", "4")), help="different read batches the steps take in turn "
                    "(step i works on batch i mod N; 10 puts configs[2]'s 10 M distinct reads through one GPU in 10 steps); at least 2")
    a = ap.parse_args()
    n_batches = max(2, a.distinct_batches)
    CB = 1                        # the batch the isolated passes, the CPU baseline and the self-check work on
    if a.print_cache_dir:
        print(os.path.join(a.index_cache, f"g{a.genome_mbp:g}_sa{a.sa_intv}_seed42_{CACHE_VERSION}") if a.index_cache else ""); return

    if a.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (torch.distributed.run) BEFORE this
        # process touches the GPU, hand their output through and leave with their exit code.  Rank 0 of the children prints the line.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE {world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    n_dev = torch.cuda.device_count()
    dev_id = local_rank % n_dev              # several ranks may share a device (BENCH_SHARE_GPU runs of the N > 1 path on one GPU)
    torch.cuda.set_device(dev_id)
    dev = torch.device("cuda", dev_id)
    L = B.load_library()
    L.bmh_set_device(dev_id)
    distributed = world > 1 or "RANK" in os.environ      # under torchrun even one rank goes through RCCL
    if distributed:
        backend = "gloo" if (n_dev < world) else "nccl"  # RCCL needs one device per rank; ranks sharing a GPU rendezvous over gloo
        dist.init_process_group(backend, rank=rank, world_size=world, **({"device_id": dev} if backend == "nccl" else {}))

    # ---------------- setup (untimed): genome + index on rank 0 -> broadcast, reads shard
    n_genome = int(a.genome_mbp * 1e6)
    t_gen = t_index = 0.0
    d = pac_t = meta = None
    # (the key names the genome generator + index layout version: a cache written by another build of either is not picked up)
    cache = a.index_cache and os.path.join(a.index_cache, f"g{a.genome_mbp:g}_sa{a.sa_intv}_seed42_{CACHE_VERSION}")

    def draw_batches(g_host, holes_):
        lo_, hi_ = shard_range(a.reads_per_gpu * world, rank, world, multiple=2 if a.paired else 1)
        out_ = []
        for bseed in [7 + 1000 * j for j in range(n_batches)]:      # --distinct-batches different batches, taken in turn step by step
            if a.paired:
                reads_, _ = B.synth.make_pairs(g_host, (hi_ - lo_) // 2, a.read_len, seed=bseed + rank, holes=holes_)
            else:
                reads_, _ = B.synth.make_reads(g_host, hi_ - lo_, a.read_len, seed=bseed + rank, holes=holes_)
            out_.append(reads_)
        return out_

    g = None
    my_reads = None
    t_reads_early = 0.0
    if rank == 0 and cache and os.path.exists(os.path.join(cache, "meta.json")):
        # setup shortcut for repeated runs on one box (the profile passes): the SAME genome + index, written by an earlier run of this command
        t0 = time.time()
        mj = json.load(open(os.path.join(cache, "meta.json")))
        ld = lambda nm, dt: torch.from_numpy(np.fromfile(os.path.join(cache, nm), dtype=dt)).to(dev)
        pac_t = ld("pac.u8", np.uint8)
        d = F.DeviceFMDIndex(mj["primary"], np.asarray(mj["L2"], np.int64), mj["seq_len"], ld("bwt.i32", np.int32), mj["sa_intv"], ld("sa.i32", np.int32), ld("bits.i32", np.int32),
                             dict(mj["stats"], loaded_from_cache=1))
        meta = {"contigs": [tuple(c) for c in mj["contigs"]], "holes": [tuple(h) for h in mj["holes"]]}
        torch.cuda.synchronize(); t_index = time.time() - t0
    elif rank == 0:
        t0 = time.time()
        g_t, meta = B.synth.make_genome_device(n_genome, dev, seed=42, return_meta=True)
        pac_t = F.pack_pac_device(g_t)
        del g_t
        torch.cuda.synchronize(); t_gen = time.time() - t0
        torch.cuda.empty_cache()
        t0 = time.time()
        d = F.build_fmd_index_device(pac_t, n_genome, sa_intv=a.sa_intv, verify=a.verify)
        t_index = time.time() - t0
        if cache:
            os.makedirs(cache, exist_ok=True)
            for nm, t in (("pac.u8", pac_t), ("bwt.i32", d.bwt_t), ("sa.i32", d.sa_t), ("bits.i32", d.bits_t)):
                t.cpu().numpy().tofile(os.path.join(cache, nm))
            json.dump({"primary": int(d.primary), "L2": [int(x) for x in d.L2], "seq_len": int(d.seq_len), "sa_intv": int(d.sa_intv), "stats": d.stats,
                       "contigs": meta["contigs"], "holes": meta["holes"]}, open(os.path.join(cache, "meta.json"), "w"))
    else:
        # the other ranks do not wait for rank 0's index with idle hands: the genome generator is deterministic (seed 42), so every rank
        # makes the same 2-bit text on its own GPU and draws its shard's reads while rank 0 builds; only the index arrays travel
        t0 = time.time()
        g_t, meta = B.synth.make_genome_device(n_genome, dev, seed=42, return_meta=True)
        pac_t = F.pack_pac_device(g_t)
        g = g_t.cpu().numpy()
        del g_t
        torch.cuda.synchronize(); t_gen = time.time() - t0
        torch.cuda.empty_cache()
        t0 = time.time()
        my_reads = draw_batches(g, meta["holes"])
        t_reads_early = time.time() - t0
    t0 = time.time()
    gloo_shared = distributed and dist.get_backend() == "gloo"
    same_text = False
    if distributed and world > 1:
        # did every rank arrive at rank 0's text?  (a 64-bit sum of the packed text; if not -- another generator build, a cache from
        # elsewhere -- the text and its metadata are broadcast like the index)
        cdev = torch.device("cpu") if gloo_shared else dev
        # (the length folded into the high bits modulo a 22-bit prime: numel << 40 left the 64 bits from 2^24 bytes of text on -- an hg38-scale text would
        # have raised here on the first run with more than one rank; found by tests/test_parallel_gpu.py's four-rank case, round 6)
        chk = pac_t.sum(dtype=torch.int64).reshape(1).to(cdev) + ((pac_t.numel() % 4194301) << 40)
        ref_chk = chk.clone(); dist.broadcast(ref_chk, 0)
        agree = (chk == ref_chk).to(torch.int64); dist.all_reduce(agree, op=dist.ReduceOp.MIN)
        same_text = bool(agree.item())
        if not same_text and rank != 0:
            pac_t = None; g = None; my_reads = None
    if gloo_shared:
        # ranks share one GPU: the broadcast goes through host memory (gloo), the arrays land in each rank's own HBM allocation
        cpu = torch.device("cpu")
        if rank == 0:
            dc = F.DeviceFMDIndex(d.primary, d.L2, d.seq_len, d.bwt_t.cpu(), d.sa_intv, d.sa_t.cpu(), d.bits_t.cpu(), d.stats)
            dc, pc, meta = broadcast_built_index(dc, pac_t.cpu(), meta, cpu, src=0, with_text=not same_text)
        else:
            dc, pc, meta = broadcast_built_index(None, None if not same_text else pac_t.cpu(), meta if same_text else None, cpu, src=0, with_text=not same_text)
            d = F.DeviceFMDIndex(dc.primary, dc.L2, dc.seq_len, dc.bwt_t.to(dev), dc.sa_intv, dc.sa_t.to(dev), dc.bits_t.to(dev), {})
            pac_t = pc.to(dev)
        del dc, pc
    else:
        d, pac_t, meta = broadcast_built_index(d, pac_t, meta, dev, src=0, with_text=not same_text)
    torch.cuda.synchronize(); t_bcast = time.time() - t0
    build_stats = d.stats
    contigs, holes = meta["contigs"], meta["holes"]
    dindex = B.Index.from_device(d.primary, d.L2.astype(np.uint64), d.seq_len, d.bwt_t, d.sa_intv, d.sa_t, d.bits_t, pac_t=pac_t, l_pac=n_genome)
    if g is None:
        g = F.unpack_pac_device(pac_t, n_genome).cpu().numpy()      # host copy of the genome: read sampling, CPU baseline, host rows
    torch.cuda.empty_cache()
    lo, hi = shard_range(a.reads_per_gpu * world, rank, world, multiple=2 if a.paired else 1)
    if my_reads is None:
        my_reads = draw_batches(g, holes)
    batches = [(reads, P.reads_to_device(reads, dev)) for reads in my_reads]
    del my_reads
    n_reads = batches[0][1].n
    from bwamem_hip.lib import ChainWorkspace, _memcpy_d2d
    params = B.ExtParams.default()
    ws = B.SeedWorkspace(n_reads, n_reads * a.read_len)
    # dry run of every batch: workspace capacities, workload statistics
    stats = []
    for reads, dr in batches:
        s = ws.seed_batch(dindex, dr.ascii, dr.offs, dr.lens, 19)
        stats.append(dict(n_seeds=int(s.n_seeds), n_smems=int(s.n_smems), n_cands=int(s.n_cands)))
    dry_order = [j for j in range(n_batches) if j != CB] + [CB]      # (the check batch last: its job lengths are copied below)
    cap_seeds = int(max(x["n_seeds"] for x in stats) * 1.25) + 4096
    cw = ChainWorkspace(n_reads, cap_seeds)
    cw.set_contigs(contigs)
    cw.set_materialize(False)             # jobs stay descriptors: the DP kernels fetch bases from the reads / 2-bit reference
    for (reads, dr), stt in ((batches[j], stats[j]) for j in dry_order):
        s = ws.seed_batch(dindex, dr.ascii, dr.offs, dr.lens, 19)
        dj = cw.chain_batch(dindex, dr.ascii, dr.offs, dr.lens, s)
        stt.update(n_jobs=int(dj.n_jobs), n_regs=int(dj.n_regs), n_heavy=int(dj.n_heavy_reads))
    n_jobs, n_regs = stats[CB]["n_jobs"], stats[CB]["n_regs"]
    jq = torch.empty(n_jobs, dtype=torch.int32, device=dev); jt = torch.empty(n_jobs, dtype=torch.int32, device=dev)
    _memcpy_d2d(jq.data_ptr(), dj.d_qlen, 4 * n_jobs); _memcpy_d2d(jt.data_ptr(), dj.d_tlen, 4 * n_jobs)
    cap_jobs = int(max(x["n_jobs"] for x in stats) * 1.25) + 4096
    cap_regs = int(max(x["n_regs"] for x in stats) * 1.25) + 4096
    out = torch.zeros(cap_jobs, 3, dtype=torch.int32, device=dev)        # (three-call form, used for the isolated kernel times)

    torch.cuda.synchronize()
    s_main = torch.cuda.Stream(device=dev)
    h_main = s_main.cuda_stream
    # how often each stage ran in this process: divides the per-process counter sums of a rocprofv3 run (scripts/summarize_profiles.py)
    passes = {"seed": 2 * len(batches), "chain": len(batches), "extend": 0}
    # ablation (BENCH_CHAIN_REPLAY=<class mask>, with --distinct-batches 2 --inflight 2: a lane then sees the same batch in every step): the chaining kernels
    # of these size classes are skipped in the timed steps and their reads' stored regions stand (knob CHAIN_REPLAY_HEAVY, csrc/chain_kernels.hip)
    chain_replay = int(os.environ.get("BENCH_CHAIN_REPLAY", "0"), 0)
    if chain_replay:
        assert n_batches == max(1, a.inflight) == 2, "BENCH_CHAIN_REPLAY needs --distinct-batches 2 --inflight 2"
    import threading
    plock = threading.Lock()

    # ---------------- batches in flight.  A batch goes through the whole path on its own stream with its own workspaces, driven by
    # its own host thread; `--inflight` of them run side by side and take the steps in turn (step i -> lane i mod N), the way the
    # reference keeps several gpu_storage batches in flight across its host threads (src/fastmap.c:417-534, gasal_gpu_storage_v).
    # The chaining of the seed-rich reads is a long chain of dependent steps on few waves: alone it leaves most of the chip idle
    # while the rest of its batch waits for it; with a second batch in flight those CUs seed and extend.
    lane_log = [] if os.environ.get("BENCH_LANE_LOG") else None

    class Lane:
        def __init__(self, k):
            self.k = k
            self.ws = ws if k == 0 else B.SeedWorkspace(n_reads, n_reads * a.read_len)
            self.cw = cw if k == 0 else ChainWorkspace(n_reads, cap_seeds)
            if k:
                self.cw.set_contigs(contigs); self.cw.set_materialize(False)
            self.stream = s_main if k == 0 else torch.cuda.Stream(device=dev)
            self.h = self.stream.cuda_stream
            self.regs = torch.zeros(cap_regs, 8, dtype=torch.int32, device=dev)
            self.out3 = torch.zeros(cap_jobs, 3, dtype=torch.int32, device=dev) if a.passes == 1 else None
            self.n_regs = 0
            self.acc = {}
            # PCIe form: the lane's own device copies of the reads (two slots: the next batch arrives while this one is worked on) and pinned host
            # buffer of the regions, its own copy streams (the DMA engines run beside the kernels) and two region buffers on the device
            self.slot = None; self.host_out = None; self.cs_in = self.cs_out = None
            self.ev_in = [None, None]; self.ev_free = None; self.ev_regs = None; self.ev_out = [None, None]; self.regs2 = None; self.n_pcie = 0

        def _h2d(self, k, host_in, i):
            """reads of step i into input slot k, on the lane's inbound copy stream, once the kernels that read the slot are through"""
            with torch.cuda.stream(self.cs_in):
                if self.ev_free is not None:
                    self.cs_in.wait_event(self.ev_free)
                for dst, src in zip(self.slot[k], host_in[i % n_batches]):
                    dst.copy_(src, non_blocking=True)
                self.ev_in[k] = torch.cuda.Event(); self.ev_in[k].record(self.cs_in)

        def step(self, i, host_in=None):
            dr = batches[i % n_batches][1]
            ascii_t, offs_t, lens_t = dr.ascii, dr.offs, dr.lens
            regs_t = self.regs
            if host_in is not None:
                if self.slot is None:
                    self.slot = [tuple(torch.empty_like(x) for x in (dr.ascii, dr.offs, dr.lens)) for _ in range(2)]
                    self.host_out = torch.empty(cap_regs, 8, dtype=torch.int32).pin_memory()
                    self.cs_in, self.cs_out = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
                    self.regs2 = [self.regs, torch.zeros_like(self.regs)]
                k = self.n_pcie & 1
                prev_done = self.ev_free                                # end of the lane's previous step: the other slot's last reader
                if self.ev_in[k] is None:                               # (the lane's first step of a loop: nothing was sent ahead)
                    self._h2d(k, host_in, i)
                self.stream.wait_event(self.ev_in[k]); self.ev_in[k] = None
                ascii_t, offs_t, lens_t = self.slot[k]
                regs_t = self.regs2[k]
                if self.ev_out[k] is not None:                          # the regions that left this buffer two steps ago have reached the host
                    self.stream.wait_event(self.ev_out[k])
            t_host = [time.time()]
            with seed_gate:
                sd = self.ws.seed_batch(dindex, ascii_t, offs_t, lens_t, 19, stream=self.h)
            t_host.append(time.time())
            tm = self.ws.timing()
            if host_in is not None and i + n_lanes < self.k_end:        # the lane's next batch sets out now, into the other slot (behind the previous step's last kernel)
                self.ev_free = prev_done
                self._h2d(k ^ 1, host_in, i + n_lanes)
            if a.passes == 1:
                dj_ = self.cw.chain_batch(dindex, ascii_t, offs_t, lens_t, sd, stream=self.h)
                self.cw.extend(self.out3, params=params, stream=self.h)
                tm["extend_one_pass"] = 0.0
                self.cw.merge(self.out3, regs_t, stream=self.h)
            else:
                with cem_gate:
                    dj_ = self.cw.extend_merge(dindex, ascii_t, offs_t, lens_t, sd, regs_t, params=params, stream=self.h)
            self.n_regs = int(dj_.n_regs); self.last_batch = i % n_batches; self.last_pcie = host_in is not None
            self.last_regs = regs_t
            t_host.append(time.time())
            if lane_log is not None:                                  # (BENCH_LANE_LOG=1: host-side begin / seeded / launched of every batch)
                lane_log.append((self.k, i, *t_host))
            if host_in is not None:                                   # regions leave over PCIe on the outbound copy stream, beside the lane's next batch
                ev = torch.cuda.Event(); ev.record(self.stream)
                with torch.cuda.stream(self.cs_out):
                    self.cs_out.wait_event(ev)
                    self.host_out[: self.n_regs].copy_(regs_t[: self.n_regs], non_blocking=True)
                    self.ev_out[k] = torch.cuda.Event(); self.ev_out[k].record(self.cs_out)
                self.n_pcie += 1
                self.ev_free = ev
            cm = self.cw.timing()
            tm["chain_light"] = cm["to_counts"]    # classify + lane kernel + counts: what the first extension waits for
            tm["chain_heavy_beside"] = cm["wave"]  # wave kernels on side streams (hidden behind extend_a as far as it lasts)
            if a.passes == 1:
                tm["extend_one_pass"] = L.bmh_extend_last_ms()
            else:
                xm = self.cw.extend_merge_timing()
                tm["extend_a"] = xm["extend_a"]; tm["extend_b"] = xm["extend_b"]; tm["chain_extend_merge"] = xm["stage"]
            for kk, v in tm.items():
                self.acc[kk] = self.acc.get(kk, 0.0) + v
            with plock:
                passes["seed"] += 1; passes["chain"] += 1; passes["extend"] += 1

    n_lanes = max(1, a.inflight)
    lanes = [Lane(k) for k in range(n_lanes)]
    # One batch in the seeding call at a time: two batches that seed side by side share the gather rate and then ask for the VALU together; with the gate they fall
    # into step -- one seeds while the other extends -- and stay there (26.8 -> 26.1 ms per step, four interleaved pairs, profiles/r05_phase_gate.txt).
    # BENCH_PHASE_LOCKS: bit 0 that gate (default), bit 1 the same around the chaining / extension call (measured: slower)
    import contextlib
    _pl = int(os.environ.get("BENCH_PHASE_LOCKS", "1"))
    seed_gate = threading.Lock() if _pl & 1 else contextlib.nullcontext()
    cem_gate = threading.Lock() if _pl & 2 else contextlib.nullcontext()

    def run_steps(k, host_in=None):
        """k steps, step i on lane i mod N; returns when all of them are through (device idle)"""
        errs = []

        def work(lane):
            try:
                L.bmh_set_device(dev_id)                              # HIP's current device is per host thread
                torch.cuda.set_device(dev_id)
                lane.k_end = k
                for i in range(lane.k, k, n_lanes):
                    lane.step(i, host_in)
            except Exception as e:                                    # noqa: BLE001 -- re-raised on the main thread
                errs.append(e)
        th = [threading.Thread(target=work, args=(ln,)) for ln in lanes[1:]]
        for t in th:
            t.start()
        work(lanes[0])
        for t in th:
            t.join()
        if errs:
            raise errs[0]
        torch.cuda.synchronize()

    run_steps(max(a.warmup, n_lanes if a.warmup else 0))
    if chain_replay:
        L.bmh_tune_set(b"CHAIN_REPLAY_HEAVY", chain_replay, 0)
    for ln in lanes:
        ln.acc = {}
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    if lane_log is not None:
        lane_log.clear()
    t0 = time.perf_counter(); t0_wall = time.time()
    run_steps(a.steps)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if lane_log:
        for k_, i_, b_, s_, e_ in sorted(lane_log, key=lambda r: r[2]):
            print(f"[lane {k_}] batch {i_}: begin {1e3 * (b_ - t0_wall):7.1f} ms  seeded {1e3 * (s_ - t0_wall):7.1f}  chain+extension launched {1e3 * (e_ - t0_wall):7.1f}", file=sys.stderr)
    stage_ms = {}
    for ln in lanes:
        for kk, v in ln.acc.items():
            stage_ms[kk] = stage_ms.get(kk, 0.0) + v
    stage_ms = {k: v / a.steps for k, v in stage_ms.items()}
    regs_out = [lanes[0].regs, lanes[-1].regs]

    # ---------------- the same steps fed over PCIe: pinned host reads -> H2D -> path -> D2H of the regions, every lane on its own stream
    # (the copies of one lane overlap the kernels of the others)
    dt_pcie = None
    if a.pcie:
        host_in = []
        for reads, dr in batches:
            host_in.append((dr.ascii.cpu().pin_memory(), dr.offs.cpu().pin_memory(), dr.lens.cpu().pin_memory()))
        run_steps(max(a.warmup, n_lanes), host_in)
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(a.steps, host_in)
        if distributed:
            dist.barrier()
        dt_pcie = time.perf_counter() - t0
        pcie_bytes = (host_in[0][0].numel() + 8 * n_reads, 32 * n_regs)

    if chain_replay:
        L.bmh_tune_set(b"CHAIN_REPLAY_HEAVY", 0, 1)
    # ---------------- the regions of the LAST TIMED step of the last lane, as they left the timed loop (over PCIe when that loop ran):
    # what the self-check below compares with the oracle.  The check is made on batch CB; a lane whose last step was another batch runs
    # one more (untimed) step of the same code path on batch CB.
    vlane = lanes[-1]
    if getattr(vlane, "last_batch", -1) != CB or chain_replay:
        vlane.k_end = 0
        vlane.step(CB, host_in if a.pcie else None)
    torch.cuda.synchronize()
    last_regs_h = vlane.host_out[: vlane.n_regs].numpy().copy() if vlane.last_pcie else vlane.last_regs[: vlane.n_regs].cpu().numpy()

    if distributed:
        tt = torch.tensor([dt, dt_pcie or 0.0], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        mine = tt[:1].clone()
        per_rank = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(per_rank, mine)
        per_rank_ms = [float(x[0]) / a.steps * 1e3 for x in per_rank]
        # what every rank did in setup: ranks other than 0 generate the text and draw their reads, then sit in the broadcast until rank 0 has
        # built (and proven) the index -- `in_broadcast` of rank r minus rank 0's is the time it waited with idle hands
        su = torch.tensor([t_gen, t_index, t_reads_early, t_bcast], dtype=torch.float64, device=tt.device)
        su_all = [torch.zeros_like(su) for _ in range(world)]
        dist.all_gather(su_all, su)
        setup_per_rank = [{"genome": round(float(x[0]), 2), "index_build": round(float(x[1]), 2), "reads_drawn_early": round(float(x[2]), 2), "in_broadcast": round(float(x[3]), 2)} for x in su_all]
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt, dt_pcie = float(tt[0]), (float(tt[1]) if a.pcie else None)
        tot = torch.tensor([n_reads], dtype=torch.int64, device=tt.device)
        dist.all_reduce(tot)
        total_reads = int(tot.item())
    else:
        total_reads = n_reads
    # per-kernel durations of one more pass on the last batch (HIP events on the launch stream), for the roofline
    iso_ms = {}
    dr = batches[CB][1]
    for _ in range(3):
        sd = ws.seed_batch(dindex, dr.ascii, dr.offs, dr.lens, 19, stream=h_main)
        tm = ws.timing()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cw.chain_batch(dindex, dr.ascii, dr.offs, dr.lens, sd, stream=h_main)
        torch.cuda.synchronize()
        tm["chain"] = (time.perf_counter() - t0) * 1e3
        cw.extend(out, params=params, stream=h_main)
        tm["extend"] = L.bmh_extend_last_ms()
        cw.merge(out, regs_out[1], stream=h_main)
        torch.cuda.synchronize()
        passes["seed"] += 1; passes["chain"] += 1; passes["extend"] += 1
        for k, v in tm.items():
            iso_ms[k] = iso_ms.get(k, 0.0) + v / 3

    # ---------------- CPU checker on a bounded sample of batch 1: the CPU baseline (N = 1) and the self-check of the timed batch (every rank)
    n_check = a.verify_sample if a.verify_sample >= 0 else (min(a.cpu_sample, n_reads) if world == 1 else min(10000, n_reads))
    n_cpu_all = min(a.cpu_sample, n_reads) if world == 1 else 0
    cb = verified = None
    t_cpu_setup = 0.0
    if max(n_check, n_cpu_all) > 0:
        from bwamem_hip.lib import SeedsT, seeds_to_host
        ncores = max(1, effective_cores() // (world if n_dev >= world else 1))
        t0 = time.time()
        hidx = F.device_index_to_host(d, max(16, a.sa_intv))
        cb = cpu_baseline(g, pac_t.cpu().numpy(), hidx, batches[CB][0], contigs, max(n_check, n_cpu_all), max(1, min(a.cpu_sample_1t, n_reads)) if world == 1 and n_cpu_all else 0, ncores)
        t_cpu_setup = time.time() - t0 - sum(cb[k][t] for k in ("all", "one") if k in cb for t in ("t_seed", "t_chain", "t_ext"))
        del hidx
        if n_check > 0:
            ck = cb["_check"]
            sd_v = ws.seed_batch(dindex, batches[CB][1].ascii, batches[CB][1].offs, batches[CB][1].lens, 19, stream=h_main)
            torch.cuda.synchronize()
            head = SeedsT.from_buffer_copy(bytes(sd_v)); head.n_seeds = min(int(sd_v.n_seeds), len(ck["seeds"]["rbeg"]))      # only the sample's seeds travel to the host
            verified = verify_against_oracle(ck, seeds_to_host(head, ck["n"]), last_regs_h)
            verified["what"] = ("GPU seeds and alignment regions of the first reads of the last timed batch (regions as they left the timed loop"
                                + (", D2H over PCIe" if vlane.last_pcie else "") + ") vs oracle seeding -> bmh_build_jobs -> oracle ksw_extend2 -> bmh_merge_regs; "
                                  "bmh_build_jobs / bmh_merge_regs are this library's HOST job builder and merge (csrc/host_jobs.cpp: product code, pinned on its own to the "
                                  "reference's recorded job stream, tests/golden/jobs_golden.npz), the seeding and the extension between them are the oracle's")
            if distributed:
                ok = torch.tensor([int(verified["seeds_identical"]), int(verified["regions_identical"]), verified["reads"]], dtype=torch.int64, device=tt.device)
                mn = ok.clone(); dist.all_reduce(mn, op=dist.ReduceOp.MIN); dist.all_reduce(ok)
                verified.update(seeds_identical=bool(mn[0]), regions_identical=bool(mn[1]), reads=int(ok[2]), ranks=world)
    failed = verified is not None and not (verified["seeds_identical"] and verified["regions_identical"])

    if rank == 0:
        ms_per_step = dt / a.steps * 1e3
        value = total_reads * a.steps / dt / 1e6
        # VALU calibration of THIS device in THIS run (bmh_calib_valu: the DP kernels' instruction kinds without their data, 8 waves per SIMD), with
        # the shader clock read inside the kernel (s_memtime over s_memrealtime around a VALU-dense loop): the extension is VALU-bound, MI355X boards
        # differ by up to 12 % in sustained clock (MI355X_MICROARCH.md), so two bench lines can be split into clock and code
        cal, clock = {}, {}
        L.bmh_calib_valu.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_double)]
        L.bmh_calib_last_clock.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double)]
        ms_c = C.c_float(0); ops_c = C.c_double(0); mhz_c = C.c_double(0); cpi_c = C.c_double(0)
        for mode, nm in ((0, "independent_v_max_add"), (1, "dependent_chain"), (2, "dpp_row_shr_max"), (4, "packed_u16_add_max")):
            if L.bmh_calib_valu(mode, 8, 20000, None, C.byref(ms_c), C.byref(ops_c)) == 0:
                cal[nm] = round(ops_c.value / (ms_c.value * 1e-3) / VALU_PEAK_LANEOPS, 3)
                if mode == 0 and L.bmh_calib_last_clock(C.byref(mhz_c), C.byref(cpi_c)) == 0:
                    clock = {"clock_mhz": round(mhz_c.value, 1), "shader_cycles_per_instr_of_the_first_wave": round(cpi_c.value, 3),
                             "how": "shader cycles (s_memtime) / 100 MHz ticks (s_memrealtime) of the first wave of the launch around its 2.56 M VALU instructions, 8 waves per SIMD "
                                    "on every SIMD (the SIMD issues its oldest ready wave first: that wave runs at the SIMD's rate, one instruction per ~4.4 cycles)"}
        st = stats[CB]
        index_how = ("loaded from --index-cache (built on the device by an earlier run of this command)" if build_stats.get("loaded_from_cache") else
                     "built on the device in setup" + (" and verified completely (every adjacent pair of suffix-array rows)" if build_stats.get("verified") else ", not verified (--no-verify-index)"))
        from bwamem_hip.lib import sources_sha16
        lib_sha = sources_sha16()
        workload_key = f"g{a.genome_mbp:g}_r{n_reads}_l{a.read_len}_{'pe' if a.paired else 'se'}_sa{a.sa_intv}"
        res = {
            "metric": "Mreads/s (150 bp single-end vs hg38-scale index; seed-and-extend hot path)", "value": round(value, 3), "unit": "Mreads/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u16", "data": "synthetic",
            "clock_mhz": clock.get("clock_mhz"), "lib_sources_sha16": lib_sha,
            "incl_pcie_value": round(total_reads * a.steps / dt_pcie / 1e6, 3) if dt_pcie else None,
            "stage_ms_isolated": {k: round(v, 3) for k, v in iso_ms.items()},
            "stage_ms": {k: round(v, 3) for k, v in stage_ms.items()},
            "verified_identical": None if verified is None else bool(verified["seeds_identical"] and verified["regions_identical"]),
            "clock": clock,
            "config": {"workload": f"{a.reads_per_gpu} synthetic {a.read_len} bp {'paired-end (interleaved)' if a.paired else 'single-end'} reads per GPU vs an hg38-scale FMD index: seeded synthetic "
                                   f"{a.genome_mbp:g} Mbp genome, 24 contigs, 50% planted repeats (mid-copy families, LINE-like, high-copy SINE-like, satellites, low-divergence segmental "
                                   f"duplications), N-runs; seq_len = {d.seq_len} rows{' > 2^32' if d.seq_len >> 32 else ''}; index {index_how}; "
                                   "seeding = all SMEMs >= 19 bp + locate; extension = every left/right job the reference's chaining (mem_chain, mem_chain_flt, mem_chain2aln) produces; "
                                   "chaining, job construction with on-device reference fetch and the region merge run on the device inside the timed region (reads in, regions out); "
                                   f"{n_batches} different read batches taken in turn ({min(n_batches, a.steps) * n_reads} distinct reads per GPU in the timed steps)",
                       "distinct_batches": n_batches, "batches_in_flight": n_lanes, "extension_passes_per_batch": a.passes, "workload_key": workload_key, "reads_per_gpu": n_reads, "read_len": a.read_len, "paired_interleaved": bool(a.paired), "genome_mbp": a.genome_mbp,
                       "seq_len": int(d.seq_len), "index_bytes": int(d.bwt_t.numel() * 4 + d.sa_t.numel() * 4 + d.bits_t.numel() * 4 + pac_t.numel()), "sa_intv": a.sa_intv,
                       "seeds_per_read": round(st["n_seeds"] / n_reads, 2), "smems_per_read": round(st["n_smems"] / n_reads, 2), "ext_jobs_per_read": round(st["n_jobs"] / n_reads, 2),
                       "regions_per_read": round(st["n_regs"] / n_reads, 2), "reads_chained_by_a_whole_wave": st["n_heavy"],
                       "ext_jobs_per_gpu": st["n_jobs"], "regions_per_gpu": st["n_regs"], "seeds_per_gpu": st["n_seeds"], "min_seed_len": 19,
                       "scoring": "a1 b4 o6 e1 clip5 zdrop0",
                       "setup_s": {"genome": round(t_gen, 2), "index_build": round(t_index, 2), "index_broadcast": round(t_bcast, 2), "reads_drawn_while_waiting": round(t_reads_early, 2)},
                       "index_build": build_stats},
            "passes": passes,
        }
        if dt_pcie:
            res["incl_pcie"] = {"value": round(total_reads * a.steps / dt_pcie / 1e6, 3), "unit": "Mreads/s", "ms_per_step": round(dt_pcie / a.steps * 1e3, 3),
                                "h2d_bytes_per_step": int(pcie_bytes[0]), "d2h_bytes_per_step": int(pcie_bytes[1]),
                                "how": "pinned host reads -> H2D -> path -> D2H of the regions into pinned host memory; every lane double-buffers both directions on copy streams "
                                       "of its own (the next batch's reads arrive while this one is worked on, its regions leave beside the next one): what a host "
                                       "worker of the reference's boundary does with two gpu_storage batches"}
        if verified is not None:
            res["verified"] = verified
        if distributed:
            res["distributed"] = {"backend": dist.get_backend(), "ranks": world, "devices_on_this_node": n_dev, "index_broadcast_s": round(t_bcast, 2),
                                  "text_generated_on_every_rank": bool(same_text), "ms_per_step_per_rank": {"min": round(min(per_rank_ms), 3), "max": round(max(per_rank_ms), 3)},
                                  "setup_s_per_rank": setup_per_rank,
                                  "ranks_idle_behind_rank0_s": round(max([0.0] + [x["in_broadcast"] - setup_per_rank[0]["in_broadcast"] for x in setup_per_rank[1:]]), 2),
                                  "what": "one process per GPU; rank 0 builds the index while the others generate the same genome and draw their shards' reads, then ONE broadcast "
                                          "of the index arrays (RCCL over xGMI with backend nccl); no collective on the data path; `value` = reads of all ranks / slowest rank's time"}
        if world > 1:
            res["note"] = "N > 1 line: `roofline` and `cpu_baseline` are reported by the N = 1 run only (rank 0 at N = 1, bench contract)"
        # ---------------- CPU baseline + roofline of the dominant kernel (N = 1 only)
        if world == 1 and n_cpu_all > 0:
            al, on = cb["all"], cb["one"]
            rate = lambda x: x["n"] / (x["t_seed"] + x.get("t_chain", 0.0) + x["t_ext"]) / 1e6
            res["cpu_baseline"] = {
                "value": round(rate(al), 5), "unit": "Mreads/s", "cores": ncores, "kind": "port",
                "sample": f"first {al['n']} reads of the last batch on {ncores} threads: oracle seeding {al['t_seed']:.2f}s + host chaining/job builder (bmh_build_jobs) "
                          f"{al['t_chain']:.2f}s + oracle extension {al['t_ext']:.2f}s",
                "cpu_model": cpu_model(), "nproc": os.cpu_count(), "usable_cores": ncores,
                "one_thread": {"value": round(rate(on), 6), "reads": on["n"], "t_seed_s": round(on["t_seed"], 3), "t_chain_s": round(on["t_chain"], 3), "t_ext_s": round(on["t_ext"], 3)},
                "all_cores": {"value": round(rate(al), 5), "reads": al["n"], "t_seed_s": round(al["t_seed"], 3), "t_chain_s": round(al["t_chain"], 3), "t_ext_s": round(al["t_ext"], 3),
                              "scaling_vs_one_thread": round(rate(al) / rate(on), 1)},
                "setup_s": round(t_cpu_setup, 1)}
            if verified is not None:
                res["cpu_baseline"]["checked_against_gpu"] = {k: verified[k] for k in ("reads", "seeds_identical", "regions_identical")}
            if "ref" in cb:
                rf = cb["ref"]
                res["cpu_baseline"]["reference_code_one_thread"] = {
                    "what": "the reference's own compiled bwt_smem1 + bwt_sa (src/bwt.c) and ksw_extend2 (src/ksw.c) from oracle/_ref/libref.so on the one-thread sample, "
                            "index re-interleaved into its 128-symbol layout; its host chaining cannot be compiled here (GASAL2 headers), the port's is used for both",
                    "t_seed_s": round(rf["t_seed"], 3), "t_ext_s": round(rf["t_ext"], 3), "identical_to_port": rf["identical_to_port"],
                    "value": round(on["n"] / (rf["t_seed"] + on["t_chain"] + rf["t_ext"]) / 1e6, 6),
                    "ref_ratio": {"seeding_port_over_ref": round(on["t_seed"] / rf["t_seed"], 2), "extension_port_over_ref": round(on["t_ext"] / rf["t_ext"], 2)}}
            res["speedup_vs_cpu_baseline"] = round(value / rate(al), 1)
            res["oracle_work_per_read"] = {k: round(v / al["n"], 2) for k, v in al["work"].items()}
            # algorithmic bytes per read, counted by the oracle on the sample
            wk, n = al["work"], al["n"]
            lf_blocks = wk["n_blk_lf"] if a.sa_intv > 1 else 0          # with every row sampled the locate kernel touches no index block
            if a.sa_intv > 1 and a.sa_intv != 16:
                lf_blocks = wk["n_blk_lf"] * (a.sa_intv - 1) / 15.0     # mean walk (intv-1)/2 instead of 7.5
            per_read = {"locate": (32.0 * lf_blocks + 4.0 * wk["n_sa"] + 20.0 * al["n_seeds"]) / n,
                        "forward": 32.0 * wk["n_blk_fwd"] / n + a.read_len / 4 + 8,
                        "backward": 32.0 * wk["n_blk_back"] / n}
            q, t = jq.long(), jt.long()
            ext_bytes = float(((q + 3) // 4 + (t + 3) // 4 + (q + t + 7) // 8 + 28).sum().item())
            kernel_bytes = {k: v * n_reads for k, v in per_read.items()}
            kernel_bytes["extend"] = ext_bytes
            cells = al["cells"] / max(al["n_jobs"], 1) * n_jobs
            prof = profile_counters(workload_key, lib_sha)
            names = {"forward": "smem_forward_kernel", "backward": "smem_backward_kernel", "locate": "locate_kernel",
                     "extend": "extension kernel family (ext_closed_form, extpk<G,P> packed 16-bit, extend16<C> / extend16_static<C> / extend_wide<C> 32-bit)", "chain": "chain_lane_kernel + chain_wave_kernel"}

            def from_prof(k, field):
                if not prof or k not in prof.get("families", {}):
                    return None
                return prof["families"][k].get(field)

            def hbm_obj(k):
                ach_ = kernel_bytes[k] / (iso_ms[k] * 1e-3) / 1e9
                o = {"bound": "hbm", "kernel": names[k], "achieved": round(ach_, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(ach_ / HBM_PEAK_GBS, 5), "traffic": from_prof(k, "hbm_bytes_per_launch"), "avg_ms": round(iso_ms[k], 3),
                     "algorithmic_bytes_per_launch": int(kernel_bytes[k])}
                if o["traffic"] is not None:
                    o["traffic_source"] = f"from_profile {prof['_file']} (same command under rocprofv3, separate --pmc passes; not measured by this run)"
                return o
            prof_src = f"from_profile {prof['_file']} (same command under rocprofv3, separate --pmc passes; not measured by this run)" if prof else None
            timed = {k: iso_ms.get(k, 0.0) for k in ("forward", "backward", "locate", "extend")}
            dom = max(timed, key=timed.get)
            hb = max((k for k in timed if k != "extend"), key=timed.get)
            res["roofline_hbm_kernel"] = hbm_obj(hb)
            # SURVEY.md 8d's figure for the path as a whole: (sum of the algorithmic bytes of seeding and extension) / step time / HBM peak
            path_bytes = float(sum(kernel_bytes.values()))
            res["roofline_path_hbm"] = {"algorithmic_bytes_per_step": int(path_bytes), "GBps": round(path_bytes / (ms_per_step * 1e-3) / 1e9, 1),
                                        "frac": round(path_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "what": "SURVEY 8d: (B_seed + B_ext) / t / HBM peak over the pipelined step; low by nature -- half of the step is integer DP that touches 110 B per job"}
            res["roofline_all"] = {k: {"ms": round(iso_ms[k], 3), "algorithmic_GBps": round(kernel_bytes[k] / (iso_ms[k] * 1e-3) / 1e9, 2)} for k in kernel_bytes}
            # the extension's binding roofline: integer VALU issue.  Algorithmic lane-ops = reference cells x 12; reference
            # cells = sum over rows of (end - beg) as ksw_extend2 executes them, counted by the oracle on the CPU sample and scaled to the batch
            t_ext_s = iso_ms["extend"] * 1e-3
            lane_ops = cells * 12.0
            valu = {"bound": "int-valu", "kernel": names["extend"], "achieved": round(lane_ops / t_ext_s / 1e12, 2), "peak": round(VALU_PEAK_LANEOPS / 1e12, 2),
                    "unit": "T lane-ops/s", "frac": round(lane_ops / t_ext_s / VALU_PEAK_LANEOPS, 4), "traffic": from_prof("extend", "hbm_bytes_per_launch"),
                    "avg_ms": round(iso_ms["extend"], 3), "algorithmic_lane_ops_per_launch": int(lane_ops), "reference_cells_per_launch": int(cells),
                    "gcups_reference_cells": round(cells / t_ext_s / 1e9, 1), "jobs_per_launch": n_jobs,
                    "peak_definition": "16-bit lane-ops: 256 CUs x 4 SIMDs x 16 lanes per clock x 2 packed halves x 2.4 GHz.  A wave64 VALU instruction occupies its SIMD for 4 cycles "
                                       "(measured: no instruction kind exceeds 0.53 of one-per-2-cycles at 1-8 waves per SIMD with every SIMD of every XCD evenly loaded, "
                                       "profiles/r04_calib_valu_placement.txt; v_pk_fma_f32 at 0.45 x 2 lanes = the chip's 157 TFLOP/s), so 78.6 T is what PACKED 16-bit "
                                       "instructions can retire and 39.3 T what 32-bit ones can; no MFMA: integer DP, not a contraction",
                    "hbm_view": {"algorithmic_bytes_per_launch": int(kernel_bytes["extend"]), "GBps": round(kernel_bytes["extend"] / t_ext_s / 1e9, 2),
                                 "frac_of_hbm_peak": round(kernel_bytes["extend"] / t_ext_s / 1e9 / HBM_PEAK_GBS, 5)}}
            if valu["traffic"] is not None:
                valu["traffic_source"] = prof_src
            valu["calibration_frac_of_peak_measured_now"] = cal
            if cal.get("independent_v_max_add"):
                # what the chip sustains on plain integer VALU instructions, measured in this run: one wave64 instruction per ~4.4 cycles per
                # SIMD whatever the number of resident waves (scripts/calib_valu.py), i.e. about half of the 2-cycle figure of the guide
                valu["measured_ceiling"] = round(cal["independent_v_max_add"] * VALU_PEAK_LANEOPS / 1e12, 2)
                valu["frac_of_measured_ceiling"] = round(valu["frac"] / cal["independent_v_max_add"], 4)
            issued = from_prof("extend", "valu_wave_instr_per_launch")
            if issued is not None:
                valu["issue_slot_frac"] = round(issued * 64.0 / t_ext_s / VALU_PEAK_LANEOPS, 4)
                valu["executed_lane_instr_per_reference_cell"] = round(issued * 64.0 / cells, 2)
                if valu.get("measured_ceiling"):
                    # how full the VALU issue port is: executed instructions against what the chip sustains on plain integer instructions --
                    # near 1 means the family is issue-bound as it stands, and only fewer instructions per reference cell can make it faster
                    valu["issue_slot_frac_of_measured_ceiling"] = round(valu["issue_slot_frac"] / cal["independent_v_max_add"], 4)
                valu["issue_source"] = "SQ_INSTS_VALU summed over the whole family, " + prof_src
            valu["dtype_note"] = "DP cells as packed unsigned 16-bit pairs (v_pk_*_u16) where h0 + qlen*a < 4096, 32-bit lanes otherwise; rank arithmetic of the seeding is 32/64-bit popcounts"
            # `roofline` = the dominant kernel family with the bound that binds it: integer VALU for the extension, HBM gathers for a seeding kernel
            res["roofline"] = valu if dom == "extend" else res["roofline_hbm_kernel"]
            res["extension_stage"] = valu
            if a.next_rows:
                try:
                    res["next_rows"] = downstream_stages(L, dindex, batches[CB][1], cw, regs_out[1], st["n_regs"], n_reads, g, pac_t, batches[CB][0], params, contigs, a.paired, reads2=batches[0][0])
                except Exception as e:                      # never lose the bench line over the extras
                    res["next_rows"] = {"error": repr(e)}
        print(json.dumps(res), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if failed:
        print("bench.py: the timed batch does NOT match the oracle: " + json.dumps(verified), file=sys.stderr)
        sys.exit(3)


def downstream_stages(L, dindex, dr, cw, regs_out, n_regs, n_reads, g, pac_t, reads, params, contigs, paired=False, reads2=None):
    """The rows after the hot path, measured on the same batch and reported beside the metric, not in it:
    bmh_finalize_regs / bmh_finalize_pairs (host threads, like the reference) and bmh_cigar_batch (device: CIGAR / NM / MD of every reported alignment)."""
    from bwamem_hip.lib import ChainOpt, PostOpt, cigar_batch, _np_ptr, _u8p, _u64p, _i32p, _u32p
    dev = regs_out.device
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    regs_h = regs_out[:n_regs].cpu().numpy()
    t_d2h = time.perf_counter() - t0
    rpr = torch.empty(n_reads, dtype=torch.int32, device=dev); fr = torch.empty(n_reads, dtype=torch.float32, device=dev)
    dj = cw.last_jobs
    B.lib._memcpy_d2d(rpr.data_ptr(), dj.d_regs_per_read, 4 * n_reads); B.lib._memcpy_d2d(fr.data_ptr(), dj.d_frac_rep, 4 * n_reads)
    rpr_h = rpr.cpu().numpy().view(np.uint32); fr_h = fr.cpu().numpy()
    co = ChainOpt(); L.bmh_chain_opt_default(C.byref(co)); po = PostOpt(); L.bmh_post_opt_default(C.byref(po))
    pac_h = pac_t.cpu().numpy()
    flat = np.ascontiguousarray(reads.reshape(-1)); rl = reads.shape[1]
    offs = np.arange(n_reads, dtype=np.uint64) * rl
    nth = effective_cores()
    ctg_len = np.ascontiguousarray([c[1] for c in contigs], dtype=np.int32)
    ctg_off = np.ascontiguousarray(np.concatenate([[0], np.cumsum(ctg_len.astype(np.int64))[:-1]]), dtype=np.int64)
    if paired:
        from bwamem_hip.lib import finalize_pairs
        buf = np.zeros((n_regs + 2 * n_reads + 1024, 16), np.int32)
        t0 = time.perf_counter()
        fin, opr, h_rec, unflag, pes = finalize_pairs(co, params, po, len(g), pac_h, flat, offs, np.full(n_reads, rl, np.uint32), regs_h, rpr_h, fr_h, contigs=contigs, n_threads=nth, out=buf)
        t_fin = time.perf_counter() - t0
        # the same with the mate rescue's local alignments as one batch on the device (bmh_finalize_pairs_dev)
        buf2 = np.zeros_like(buf)
        t0 = time.perf_counter()
        fin2, opr2, h2, uf2, _ = finalize_pairs(co, params, po, len(g), pac_h, flat, offs, np.full(n_reads, rl, np.uint32), regs_h, rpr_h, fr_h, contigs=contigs, n_threads=nth, out=buf2,
                                                device=(dindex, dr.ascii, dr.offs, None))
        t_fin2 = time.perf_counter() - t0
        same = bool(np.array_equal(fin, fin2) and np.array_equal(opr, opr2) and np.array_equal(h_rec, h2) and np.array_equal(unflag, uf2))
        need = np.zeros(max(len(fin), 1), np.uint8)
        fin = np.ascontiguousarray(fin)
        L.bmh_sam_need_cigar_pe(C.byref(po), _np_ptr(fin, _i32p), _np_ptr(np.ascontiguousarray(opr), _u32p), _np_ptr(np.ascontiguousarray(h_rec), _i32p), n_reads, _np_ptr(need, _u8p))
        sel = np.nonzero(need[: len(fin)])[0].astype(np.int32)
        name, extra = "finalize_pairs_host", {"insert_size_FR": {"low": pes[1][0], "high": pes[1][1], "mean": round(pes[1][3], 1), "sd": round(pes[1][4], 1)},
                                               "with_mate_rescue_on_the_device": {"ms": round(t_fin2 * 1e3, 2), "identical": same,
                                                                                  "what": "bmh_finalize_pairs_dev: the ksw_align2 calls of mem_matesw as one batch of 16-lane jobs on the device, the rest on host threads"}}
        m = len(fin)
    else:
        fin = np.zeros((max(n_regs, 1), 16), np.int32); opr = np.zeros(n_reads, np.uint32)
        t0 = time.perf_counter()
        m = L.bmh_finalize_regs(C.byref(co), C.byref(params), C.byref(po), len(g), _np_ptr(pac_h, _u8p), n_reads, _np_ptr(flat, _u8p), _np_ptr(offs, _u64p),
                                _np_ptr(np.ascontiguousarray(regs_h), _i32p), _np_ptr(np.ascontiguousarray(rpr_h), _u32p), fr_h.ctypes.data_as(C.POINTER(C.c_float)),
                                len(contigs), ctg_off.ctypes.data_as(C.c_void_p), _np_ptr(fin, _i32p), _np_ptr(opr, _u32p), nth)
        t_fin = time.perf_counter() - t0
        if m < 0:
            raise RuntimeError("bmh_finalize_regs failed")
        fin = fin[:m]
        # the same tail on the device (csrc/regs_kernels.hip), on the regions where the merge kernel left them
        from bwamem_hip.lib import finalize_regs_device
        d_out = torch.empty(max(n_regs, 1), 16, dtype=torch.int32, device=dev); d_opr = torch.empty(n_reads, dtype=torch.int32, device=dev)
        ms_dev = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            fd, _o = finalize_regs_device(dindex, co, params, po, dr.ascii, dr.offs, regs_out, n_regs, dj.d_regs_per_read, dj.d_frac_rep, n_reads, contigs=contigs, out_t=d_out, opr_t=d_opr)
            ms_dev.append(((time.perf_counter() - t0) * 1e3, L.bmh_finalize_regs_device_last_ms()))
        same = bool(fd.shape[0] == m and torch.equal(fd.cpu(), torch.from_numpy(fin)) and np.array_equal(d_opr.cpu().numpy().view(np.uint32), opr))
        sel = np.nonzero(fin[:, 15])[0].astype(np.int32)
        name, extra = "finalize_regs_host", {"reported": int(len(sel))}
        dev_row = {"finalize_regs_device": {"ms": round(min(x[0] for x in ms_dev), 3), "kernel_ms": round(min(x[1] for x in ms_dev), 3), "regions_in": int(n_regs), "regions_out": int(fd.shape[0]),
                                            "identical_to_host_form": same, "what": "mem_sort_dedup_patch + mem_mark_primary_se + mem_approx_mapq_se + the selection of mem_reg2sam on the device "
                                            "(lane per read, wave per read for reads with more than 8 regions), regions in HBM -> records in HBM"}}
    out_t = torch.from_numpy(fin.copy()).to(dev); sel_t = torch.from_numpy(sel).to(dev)
    ms = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        cg, aln, md = cigar_batch(dindex, dr.ascii, dr.offs, dr.lens, out_t, len(sel), sel_t=sel_t, params=params, max_cigar=24, md_cap=128)
        torch.cuda.synchronize(); ms.append((time.perf_counter() - t0) * 1e3)
    # reads in host memory -> SAM text through the native pipeline (bmh_aligner_run: batches driven by C threads): the whole gase_aln run on FOUR batches of
    # the step's size (the step's two batches, twice: a batch of a quarter of a million reads spends a third of its time in fixed per-batch latencies,
    # and the reference's own batches are 10 Mbases per thread), two lanes; second of two runs (the first one allocates lanes, workspaces and pinned buffers)
    sam_row = {}
    try:
        from bwamem_hip.aligner import ReadSet
        from bwamem_hip.lib import NativeAligner, PeOpt
        pe_o = PeOpt(); L.bmh_pe_opt_default(C.byref(pe_o))
        flat2 = flat if reads2 is None else np.ascontiguousarray(reads2.reshape(-1))
        flat4 = np.concatenate([flat, flat2, flat, flat2])
        n4 = 4 * n_reads
        asc = B.synth.codes_to_ascii(flat4)
        w = len(str(n4))
        names = np.char.add("r", np.char.zfill((np.arange(n4) // (2 if paired else 1)).astype(str), w))
        blob = np.frombuffer(("\0".join(names.tolist()) + "\0").encode(), dtype=np.uint8)
        noff = np.arange(n4, dtype=np.uint64) * np.uint64(w + 2)
        rs = ReadSet(asc, np.arange(n4, dtype=np.uint64) * np.uint64(rl), np.full(n4, rl, np.uint32), blob, noff, codes=flat4)
        # single-end: four batches of a million reads on two lanes; interleaved pairs (the host walks of mem_sam_pe between the device stages): eight batches on four
        lanes_n = int(os.environ.get("BENCH_SAM_LANES", "4" if paired else "2"))
        nb4 = int(os.environ.get("BENCH_SAM_BATCHES", "8" if paired else "4"))
        cuts4 = [((n4 * k // nb4) & ~1) for k in range(nb4)] + [n4]
        nat = NativeAligner(dindex, pac_h, len(g), contigs, None, co, params, po, pe_o)
        nbytes = [0]
        # the letters in REGISTERED host memory (bmh_host_pin = hipHostRegister, once, before the runs): the batches go to the device straight from the caller's
        # buffer, as the batches of a read file go from the loader's pinned buffers; BENCH_SAM_PIN=0: pageable, staged by the lanes' host threads as until round 5
        L.bmh_host_pin.argtypes = [C.c_void_p, C.c_size_t]; L.bmh_host_unpin.argtypes = [C.c_void_p]
        pinned_reads = os.environ.get("BENCH_SAM_PIN", "1") != "0" and L.bmh_host_pin(asc.ctypes.data, asc.nbytes) == 0

        def sink(mv):
            nbytes[0] += len(mv)

        def cpu_ms():
            try:
                return int(dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))["usage_usec"]) / 1e3
            except Exception:               # noqa: BLE001
                return float("nan")
        # one run to warm the lanes, then five: the median is the row (a run is 0.2-0.3 s and the device is shared by the lanes' kernels in an order that differs
        # from run to run: single runs scatter by +-5 %; every run is listed)
        st_n = None
        runs_n = []
        for it in range(6):
            nbytes[0] = 0
            c0 = cpu_ms()
            st_i = nat.run(rs, cuts4, paired, sink, n_lanes=lanes_n, n_threads=nth)
            c1_i = cpu_ms()
            if it:
                runs_n.append((st_i.seconds, st_i, c1_i - c0))
        runs_n.sort(key=lambda t: t[0])
        _, st_n, cpu_used = runs_n[len(runs_n) // 2]
        c0, c1 = 0.0, cpu_used
        # ... and the same reads from a FILE (bmh_aligner_run_fasta: a loader thread cuts and fills the batches of the mapped file ahead of the lanes), the file in the
        # page cache as a freshly written read file would be: what `bwa mem ref.fa reads.fa` starts from
        file_row = {}
        try:
            import tempfile
            fa = os.path.join(tempfile.gettempdir(), "bmh_bench_reads_%d.fa" % os.getpid())
            recs = np.empty((n4, w + 3 + rl + 1), np.uint8)
            recs[:, 0] = ord(">"); recs[:, 1:w + 2] = np.frombuffer("".join(names.tolist()).encode(), np.uint8).reshape(n4, w + 1); recs[:, w + 2] = 10
            recs[:, w + 3:w + 3 + rl] = asc.reshape(n4, rl); recs[:, -1] = 10
            recs.tofile(fa); del recs
            runs_f = []
            for it in range(4):
                nbytes_f = [0]
                t0f = time.perf_counter()
                st_f = nat.run_fasta(fa, paired, lambda mv: nbytes_f.__setitem__(0, nbytes_f[0] + len(mv)), batch_reads=(n4 // nb4) & ~1, n_lanes=lanes_n, n_threads=nth)
                if it:
                    runs_f.append(time.perf_counter() - t0f)
            dt_f = sorted(runs_f)[len(runs_f) // 2]
            os.remove(fa)
            file_row = {"file_to_sam_native": {"Mreads_per_s": round(n4 / dt_f / 1e6, 2), "ms": round(dt_f * 1e3, 1), "reads": int(n4), "sam_bytes": int(nbytes_f[0]), "identical_text_size": bool(nbytes_f[0] == nbytes[0]), "runs_ms": [round(v * 1e3, 1) for v in runs_f], "row_is": "median",
                                               "what": "bmh_aligner_run_fasta: a FASTA file (in the page cache) -> batches cut and filled by a loader thread -> the same pipeline -> SAM text at the sink"}}
        except Exception as e:                              # noqa: BLE001
            file_row = {"file_to_sam_native": {"error": repr(e)}}
        nat.free()
        if pinned_reads:
            L.bmh_host_unpin(asc.ctypes.data)
        gbs = lambda b, t: round(b / t / 1e9, 1) if t > 0 else None
        sam_row = {**file_row, "reads_to_sam_native": {"Mreads_per_s": round(n4 / st_n.seconds / 1e6, 2), "ms": round(st_n.seconds * 1e3, 1), "reads": int(n4), "sam_bytes": int(nbytes[0]), "batches": nb4, "lanes": lanes_n,
                                           "runs_ms": [round(t[0] * 1e3, 1) for t in runs_n], "row_is": "median",
                                           "host_cpu_ms_per_million_reads": round((c1 - c0) / (n4 / 1e6), 1), "writer_format_ms": round(st_n.format_seconds * 1e3, 1),
                                           "lanes_ms_summed": {"h2d": round(st_n.h2d_seconds * 1e3, 1), "seeding": round(st_n.seed_seconds * 1e3, 1), "chain_extend_merge": round(st_n.chain_extend_seconds * 1e3, 1),
                                                               "tail": round(st_n.tail_seconds * 1e3, 1), "select": round(st_n.select_seconds * 1e3, 1), "cigar_text_d2h": round(st_n.cigar_seconds * 1e3, 1),
                                                               "wait_for_a_device_slot": round(st_n.gate_wait_seconds * 1e3, 1)},
                                           "copies": {"h2d_bytes": int(st_n.h2d_bytes), "h2d_ms": round(st_n.h2d_copy_seconds * 1e3, 2), "h2d_GBps": gbs(st_n.h2d_bytes, st_n.h2d_copy_seconds),
                                                      "d2h_bytes": int(st_n.d2h_bytes), "d2h_ms": round(st_n.d2h_copy_seconds * 1e3, 2), "d2h_GBps": gbs(st_n.d2h_bytes, st_n.d2h_copy_seconds),
                                                      "reads_in_registered_host_memory": bool(pinned_reads),
                                                      "how": "HIP events on the lanes' streams around the copies themselves (reads + offsets + names in, SAM text out; the lanes' copies share the link "
                                                             "with each other).  lanes_ms_summed.h2d / .cigar_text_d2h are HOST clocks around whole stages -- the staging of offsets and names on host "
                                                             "threads, the CIGAR and text kernels the host waits for -- not copy times"},
                                           "what": "bmh_aligner_run: ASCII reads in host memory -> H2D -> seeding -> chaining -> extension -> merge -> region tail -> selection of the records -> "
                                                   "CIGAR / NM / MD -> the SAM text written on the device (bmh_sam_text_*; interleaved pairs: the region tail of mem_sam_pe on host threads in between) -> "
                                                   "D2H of the text -> a sink that counts it; the same text the golden SAM tests compare with the reference's"}}
    except Exception as e:                                  # noqa: BLE001 -- an extra, never at the expense of the line
        sam_row = {"reads_to_sam_native": {"error": repr(e)}}
    return {**(dev_row if not paired else {}), **sam_row, name: dict({"ms": round(t_fin * 1e3, 2), "threads": nth, "regions_in": int(n_regs), "regions_out": int(m), "d2h_regions_ms": round(t_d2h * 1e3, 2)}, **extra),
            "cigar_batch_device": {"ms": round(min(ms), 3), "alignments": int(len(sel)), "M_alignments_per_s": round(len(sel) / (min(ms) * 1e-3) / 1e6, 1),
                                   "flagged": int((aln[:, 7].cpu().numpy() & ~2 != 0).sum())}}


if __name__ == "__main__":
    main()
