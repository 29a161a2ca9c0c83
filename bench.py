#!/usr/bin/env python3
"""bench.py -- seed-and-extend hot path on MI355X: Mreads/s, roofline and CPU baseline.

Contract (driver): python bench.py --gpus N --steps K --warmup W ; for N > 1 launched by
torch.distributed.run, one rank per GPU.  Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[1], the configuration the metric is quoted on):
  1 M synthetic 150 bp single-end reads per GPU against a seeded synthetic genome standing in
  for hg38 (no network, no hg38 on the box; see DESIGN.md "workload").  A step = one pass of the
  hot path over the batch, reads in -> alignment regions out, entirely on the device:
  SMEM seeding (pack, forward, backward, filter, expand, locate kernels) -> chaining, chain filter and
  extension-job construction with on-device reference fetch (bmh_chain_batch) -> seed extension
  (ksw_extend2 kernels) -> region merge.  Only reads and index are resident in HBM when the timed
  region starts; nothing of the batch is prepared on the host.  (--host-jobs = the earlier mode:
  jobs prebuilt by the host job builder outside the timed region, seeding || extension.)
Multi-GPU: reads shard across ranks (weak scaling: READS_PER_GPU per rank); rank 0 builds the
index and broadcasts it over RCCL once, outside the timed region; no data-path collective.
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
from bwamem_hip import pipeline as P  # noqa: E402
from bwamem_hip.parallel import broadcast_index, shard_range  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# measured ceiling of random 32-byte block gathers on this chip (scripts/calib.py, bmh_calib_gather):
# 56.8 G gathers/s = 1818 GB/s of useful bytes (each gather moves one 64-byte sector: 3.6 TB/s of traffic)
GATHER_CEILING_GBS = 1818.0


def pmc_traffic(kernel_name: str):
    """HBM bytes per launch of a kernel from the committed rocprofv3 PMC passes of this same command
    (profiles/r01f_pmc_fetch_write.json: FETCH_SIZE + WRITE_SIZE, KB).  For this library's access patterns
    FETCH_SIZE needs no correction: the calibration kernel (bmh_calib_gather under --pmc FETCH_SIZE) reads back
    63.9 B per 32-byte gather, i.e. exactly one 64-byte sector each.  None if the profile is absent."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r01f_pmc_fetch_write.json")))
        key = kernel_name.split("<")[0].split(" ")[0]
        f = [v["avg_per_launch_KB"] for k, v in d["FETCH_SIZE"].items() if key in k]
        w = [v["avg_per_launch_KB"] for k, v in d["WRITE_SIZE"].items() if key in k]
        if not f:
            return None
        return int((sum(f) + sum(w)) * 1024)
    except Exception:
        return None


def pmc_valu_busy(pred):
    """VALU-busy fraction of a kernel family from the committed SQ counter pass (profiles/r01f_pmc_sq.json):
    SQ_ACTIVE_INST_VALU counts quad-cycles per SIMD, GRBM_GUI_ACTIVE cycles summed over the 8 XCDs (MI355X_MICROARCH.md),
    so busy = 4 * sum(ACTIVE_INST_VALU) / (1024 SIMDs * sum(GUI_ACTIVE) / 8).  None if the profile is absent."""
    try:
        sq = json.load(open(os.path.join(ROOT, "profiles", "r01f_pmc_sq.json")))
        a = g = 0.0
        for k, v in sq.items():
            if pred(k) and "SQ_ACTIVE_INST_VALU" in v and "GRBM_GUI_ACTIVE" in v:
                a += v["SQ_ACTIVE_INST_VALU"] * v["launches"]; g += v["GRBM_GUI_ACTIVE"] * v["launches"]
        return round(4.0 * a / (1024.0 * g / 8.0), 4) if g else None
    except Exception:
        return None


def downstream_stages(L, dindex, dr, cw, regs_out, n_regs, n_reads, g, pac_t, reads, params, paired=False):
    """The rows after the hot path, measured on the same batch and reported beside the metric, not in it:
    bmh_finalize_regs (host: sort/dedup/patch, primary marking, MAPQ, selection -- the reference runs it on host threads too)
    and bmh_cigar_batch (device: CIGAR / NM / MD of every reported alignment)."""
    from bwamem_hip.lib import ChainOpt, PostOpt, cigar_batch, dev_jobs_to_host, _np_ptr, _u8p, _u64p, _i32p, _u32p
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
    out = np.zeros((max(n_regs, 1), 16), np.int32); opr = np.zeros(n_reads, np.uint32)
    nth = os.cpu_count() or 1
    if paired:
        from bwamem_hip.lib import finalize_pairs
        buf = np.zeros((n_regs + 2 * n_reads + 1024, 16), np.int32)             # like `out` of the single-end call: allocated (and touched) outside the timed call
        t0 = time.perf_counter()
        fin, opr, h_rec, unflag, pes = finalize_pairs(co, params, po, len(g), pac_h, flat, offs, np.full(n_reads, rl, np.uint32), regs_h, rpr_h, fr_h, n_threads=nth, out=buf)
        t_fin = time.perf_counter() - t0
        need = np.zeros(max(len(fin), 1), np.uint8)
        fin = np.ascontiguousarray(fin)
        L.bmh_sam_need_cigar_pe(C.byref(po), _np_ptr(fin, _i32p), _np_ptr(np.ascontiguousarray(opr), _u32p), _np_ptr(np.ascontiguousarray(h_rec), _i32p), n_reads,
                                _np_ptr(need, _u8p))
        sel = np.nonzero(need[: len(fin)])[0].astype(np.int32)
        out_t = torch.from_numpy(fin.copy()).to(dev); sel_t = torch.from_numpy(sel).to(dev)
        ms = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            cg, aln, md = cigar_batch(dindex, dr.ascii, dr.offs, dr.lens, out_t, len(sel), sel_t=sel_t, params=params, max_cigar=24, md_cap=128)
            torch.cuda.synchronize(); ms.append((time.perf_counter() - t0) * 1e3)
        return {"finalize_pairs_host": {"ms": round(t_fin * 1e3, 2), "threads": nth, "regions_in": int(n_regs), "regions_out": int(len(fin)),
                                        "insert_size_FR": {"low": pes[1][0], "high": pes[1][1], "mean": round(pes[1][3], 1), "sd": round(pes[1][4], 1)},
                                        "d2h_regions_ms": round(t_d2h * 1e3, 2)},
                "cigar_batch_device": {"ms": round(min(ms), 3), "alignments": int(len(sel)), "M_alignments_per_s": round(len(sel) / (min(ms) * 1e-3) / 1e6, 1),
                                       "flagged": int((aln[:, 7].cpu().numpy() & ~2 != 0).sum())}}
    t0 = time.perf_counter()
    m = L.bmh_finalize_regs(C.byref(co), C.byref(params), C.byref(po), len(g), _np_ptr(pac_h, _u8p), n_reads, _np_ptr(flat, _u8p), _np_ptr(offs, _u64p),
                            _np_ptr(np.ascontiguousarray(regs_h), _i32p), _np_ptr(np.ascontiguousarray(rpr_h), _u32p), fr_h.ctypes.data_as(C.POINTER(C.c_float)), 1, None,
                            _np_ptr(out, _i32p), _np_ptr(opr, _u32p), nth)
    t_fin = time.perf_counter() - t0
    if m < 0:
        raise RuntimeError("bmh_finalize_regs failed")
    out = out[:m]
    sel = np.nonzero(out[:, 15])[0].astype(np.int32)
    out_t = torch.from_numpy(out.copy()).to(dev); sel_t = torch.from_numpy(sel).to(dev)
    ms = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        cg, aln, md = cigar_batch(dindex, dr.ascii, dr.offs, dr.lens, out_t, len(sel), sel_t=sel_t, params=params, max_cigar=24, md_cap=128)
        torch.cuda.synchronize(); ms.append((time.perf_counter() - t0) * 1e3)
    fl = aln[:, 7].cpu().numpy()
    return {"finalize_regs_host": {"ms": round(t_fin * 1e3, 2), "threads": nth, "regions_in": int(n_regs), "regions_out": int(m), "reported": int(len(sel)),
                                   "d2h_regions_ms": round(t_d2h * 1e3, 2)},
            "cigar_batch_device": {"ms": round(min(ms), 3), "alignments": int(len(sel)), "M_alignments_per_s": round(len(sel) / (min(ms) * 1e-3) / 1e6, 1),
                                   "flagged": int((fl != 0).sum())}}


def cpu_baseline(g, idx, reads, sample: int, n_threads: int):
    """Oracle (our C restatement of the reference CPU path, parity-pinned to the compiled
    reference) timed on the host cores on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_py
    orc = oracle_py.Oracle()
    f = orc.fmd(idx)
    sub = reads[:sample]
    L = sub.shape[1]
    flat = sub.reshape(-1)
    offs = np.arange(sub.shape[0], dtype=np.uint64) * L
    lens = np.full(sub.shape[0], L, np.uint32)
    t0 = time.time()
    s = orc.seed_reads(f, flat, offs, lens, 19, n_threads=n_threads)
    t_seed = time.time() - t0
    # the same extension jobs as the GPU leg (host job builder on the oracle's seeds; untimed on both legs)
    from bwamem_hip.lib import HostJobs
    t0 = time.time()
    hj = HostJobs(g, flat, offs, lens, s, n_threads=n_threads)
    t_chain = time.time() - t0
    arr = [x.copy() for x in hj.jobs()]
    n_jobs = hj.n_jobs
    hj.free()
    t0 = time.time()
    _, _, cells = orc.extend_batch(*arr, n_threads=n_threads)
    t_ext = time.time() - t0
    return dict(t_seed=t_seed, t_ext=t_ext, t_chain=t_chain, n=sub.shape[0], work=s["work"], cells=cells, n_jobs=n_jobs,
                n_seeds=int(len(s["rbeg"])))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--genome-mbp", type=float, default=float(os.environ.get("BENCH_GENOME_MBP", "1000")))
    ap.add_argument("--reads-per-gpu", type=int, default=int(os.environ.get("BENCH_READS_PER_GPU", "1000000")))
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--sa-intv", type=int, default=1, help="suffix-array samples resident in HBM: every N-th row (16 = what the reference's `bwa index` "
                    "writes for its GPU index, src/bwtindex.c:324; the index is built that way and bmh_index_densify_sa fills in the rest on "
                    "the device); 1 = the whole suffix array, locating a seed is one gather")
    ap.add_argument("--paired", action="store_true", help="interleaved 2 x read-len pairs (configs[3]); reads-per-gpu counts reads, shards stay on pair boundaries")
    ap.add_argument("--no-overlap", dest="overlap", action="store_false", help="(--host-jobs) run extension and seeding on one stream")
    ap.add_argument("--host-jobs", action="store_true", help="round-1 mode: extension jobs prebuilt by the host job builder outside the timed region; "
                    "default: the whole path reads -> seeds -> chains/jobs -> extension -> regions runs on the device inside the timed region")
    ap.add_argument("--cpu-sample", type=int, default=int(os.environ.get("BENCH_CPU_SAMPLE", "200000")))
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == a.gpus or world == 1, f"--gpus {a.gpus} but WORLD_SIZE {world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    B.load_library().bmh_set_device(local_rank)
    distributed = world > 1 or "RANK" in os.environ      # under torchrun even one rank goes through RCCL
    if distributed:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # ---------------- setup (untimed): genome, index (rank 0) -> RCCL broadcast, reads shard, jobs
    n_genome = int(a.genome_mbp * 1e6)
    g = B.synth.make_genome(n_genome, seed=42)            # every rank regenerates the same genome
    idx = None
    t0 = time.time()
    if rank == 0:
        idx = B.fmindex.build_fmd_index(g, sa_intv=16, device=str(dev))     # the reference's sampling (src/bwtindex.c:324); densified below
    t_index = time.time() - t0
    torch.cuda.empty_cache()
    hdr, bwt_t, sa_t, bits_t = broadcast_index(idx, dev, src=0, world=world)
    # 2-bit forward strand (the .pac body) for the on-device reference fetch; every rank packs its own copy
    gp = torch.from_numpy(np.concatenate([g, np.zeros((-len(g)) % 4 + 64, np.uint8)])).to(dev).view(-1, 4).to(torch.int32)
    pac_t = ((gp[:, 0] << 6) | (gp[:, 1] << 4) | (gp[:, 2] << 2) | gp[:, 3]).to(torch.uint8).contiguous()
    del gp
    dindex = B.Index.from_device(hdr["primary"], hdr["L2"], hdr["seq_len"], bwt_t, hdr["sa_intv"], sa_t, bits_t, pac_t=pac_t, l_pac=len(g))
    t0 = time.time()
    dindex.densify_sa(a.sa_intv)                 # every rank fills in its own denser samples (device, untimed setup like the index load)
    torch.cuda.synchronize()
    t_densify = time.time() - t0
    lo, hi = shard_range(a.reads_per_gpu * world, rank, world, multiple=2 if a.paired else 1)
    if a.paired:
        reads, _ = B.synth.make_pairs(g, (hi - lo) // 2, a.read_len, seed=7 + rank)
    else:
        reads, _ = B.synth.make_reads(g, hi - lo, a.read_len, seed=7 + rank)
    dr = P.reads_to_device(reads, dev)
    n_reads = dr.n
    ws = B.SeedWorkspace(n_reads, n_reads * a.read_len)
    s = ws.seed_batch(dindex, dr.ascii, dr.offs, dr.lens, 19)
    from bwamem_hip.lib import ChainWorkspace, HostJobs, seeds_to_host
    params = B.ExtParams.default()
    L = B.load_library()
    t_jobs = 0.0
    if a.host_jobs:
        # extension jobs of the batch from the host job builder (chain -> chain_flt -> chain2aln restatement, parity-checked
        # against the reference's own host code) on all host cores, then uploaded; untimed, like the reference's host stage
        t0 = time.time()
        flat = reads.reshape(-1)
        hj = HostJobs(g, flat, np.arange(n_reads, dtype=np.uint64) * a.read_len, np.full(n_reads, a.read_len, np.uint32), seeds_to_host(s, n_reads))
        t_jobs = time.time() - t0
        jobs = P.ExtJobs(*[torch.from_numpy(np.ascontiguousarray(x).view(np.int32) if x.dtype == np.uint32 else np.ascontiguousarray(x)).to(dev)
                           for x in hj.jobs()], torch.from_numpy(hj.job_read.view(np.int32).copy()).to(dev), torch.from_numpy(hj.job_side.view(np.int32).copy()).to(dev))
        n_regs, n_jobs = hj.n_regs, jobs.n
        hj.free()
        out = torch.zeros(max(n_jobs, 1), 3, dtype=torch.int32, device=dev)
    else:
        # the device job builder: nothing of the batch is prepared on the host
        cw = ChainWorkspace(n_reads, int(s.n_seeds * 1.25) + 4096)
        cw.set_materialize(False)             # jobs stay descriptors: the DP kernels fetch bases from the reads / 2-bit reference
        dj = cw.chain_batch(dindex, dr.ascii, dr.offs, dr.lens, s)
        n_regs, n_jobs = int(dj.n_regs), int(dj.n_jobs)
        out = torch.zeros(int(n_jobs * 1.25) + 4096, 3, dtype=torch.int32, device=dev)
        regs_out = torch.zeros(int(n_regs * 1.25) + 4096, 8, dtype=torch.int32, device=dev)
        jq = torch.empty(n_jobs, dtype=torch.int32, device=dev); jt = torch.empty(n_jobs, dtype=torch.int32, device=dev)
        from bwamem_hip.lib import _memcpy_d2d
        _memcpy_d2d(jq.data_ptr(), dj.d_qlen, 4 * n_jobs); _memcpy_d2d(jt.data_ptr(), dj.d_tlen, 4 * n_jobs)
        jobs = None

    torch.cuda.synchronize()
    s_seed, s_ext = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    h_seed = s_seed.cuda_stream
    h_ext = s_ext.cuda_stream if (a.overlap and a.host_jobs) else h_seed
    chain_ms = [0.0]

    if a.host_jobs:
        # Two HIP streams: the extension of a batch runs beside the seeding of a batch.
        def step():
            B.extend_batch(jobs.q, jobs.qoff, jobs.qlen, jobs.t, jobs.toff, jobs.tlen, jobs.h0, out, params=params, stream=h_ext)
            ws.seed_batch(dindex, dr.ascii, dr.offs, dr.lens, 19, stream=h_seed)
    else:
        # reads -> seeds -> chains / jobs (incl. reference fetch) -> extension -> regions, one stream, all in HBM
        def step():
            sd = ws.seed_batch(dindex, dr.ascii, dr.offs, dr.lens, 19, stream=h_seed)
            t0 = time.perf_counter()
            d = cw.chain_batch(dindex, dr.ascii, dr.offs, dr.lens, sd, stream=h_seed)
            chain_ms[0] = (time.perf_counter() - t0) * 1e3
            cw.extend(out, params=params, stream=h_seed)
            cw.merge(out, regs_out, stream=h_seed)

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stage_ms = {}
    for _ in range(a.steps):
        step()
        tm = ws.timing()                       # HIP events on the launch stream, per stage
        tm["extend"] = L.bmh_extend_last_ms()  # idem for the DP kernels (waits for them)
        if not a.host_jobs:
            tm["chain"] = chain_ms[0]          # host clock around bmh_chain_batch (it synchronises; the base fetch kernel trails it)
        for k, v in tm.items():
            stage_ms[k] = stage_ms.get(k, 0.0) + v
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if distributed:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        tot = torch.tensor([n_reads], dtype=torch.int64, device=dev)
        dist.all_reduce(tot)
        total_reads = int(tot.item())
    else:
        total_reads = n_reads
    stage_ms = {k: v / a.steps for k, v in stage_ms.items()}
    # per-kernel durations without inter-stream interference (HIP events on the launch stream), for the roofline
    iso_ms = {}
    for _ in range(3):
        sd = ws.seed_batch(dindex, dr.ascii, dr.offs, dr.lens, 19, stream=h_seed)
        tm = ws.timing()
        torch.cuda.synchronize()
        if a.host_jobs:
            B.extend_batch(jobs.q, jobs.qoff, jobs.qlen, jobs.t, jobs.toff, jobs.tlen, jobs.h0, out, params=params, stream=h_seed)
        else:
            t0 = time.perf_counter()
            d = cw.chain_batch(dindex, dr.ascii, dr.offs, dr.lens, sd, stream=h_seed)
            torch.cuda.synchronize()
            tm["chain"] = (time.perf_counter() - t0) * 1e3
            cw.extend(out, params=params, stream=h_seed)
        tm["extend"] = L.bmh_extend_last_ms()
        torch.cuda.synchronize()
        for k, v in tm.items():
            iso_ms[k] = iso_ms.get(k, 0.0) + v / 3

    if rank == 0:
        ms_per_step = dt / a.steps * 1e3
        value = total_reads * a.steps / dt / 1e6
        res = {
            "metric": "Mreads/s (150 bp single-end seed-and-extend hot path)", "value": round(value, 3), "unit": "Mreads/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": f"{a.reads_per_gpu} synthetic {a.read_len} bp single-end reads per GPU vs seeded synthetic "
                                   f"{a.genome_mbp:g} Mbp genome (hg38 stand-in: uniform + 10% diverged repeat families); "
                                   "seeding = all SMEMs >= 19 bp + locate; extension = every left/right job the reference's chaining (mem_chain, mem_chain_flt, mem_chain2aln) produces; "
                                   + ("jobs prebuilt on the host outside the timed region" if a.host_jobs else
                                      "chaining, job construction with on-device reference fetch and the region merge run on the device inside the timed region (reads in, regions out)"),
                       "reads_per_gpu": n_reads, "read_len": a.read_len, "paired_interleaved": bool(a.paired), "genome_mbp": a.genome_mbp,
                       "index_bytes": int(bwt_t.numel() * 4 + (hdr["seq_len"] // a.sa_intv + 1) * 4.125),
                       "sa_intv": a.sa_intv,      # suffix-array samples of every sa_intv-th row resident (reference files: 16; bmh_index_densify_sa fills in)
                       "sa_densify_s": round(t_densify, 3),
                       "ext_jobs_per_gpu": n_jobs, "regions_per_gpu": n_regs, "job_builder": "host (untimed)" if a.host_jobs else "device (timed)", "host_job_build_s": round(t_jobs, 2), "seeds_per_gpu": int(s.n_seeds), "min_seed_len": 19,
                       "scoring": "a1 b4 o6 e1 clip5 zdrop0", "streams": "seeding || extension" if (a.overlap and a.host_jobs) else "single", "index_build_s": round(t_index, 2)},
            "stage_ms": {k: round(v, 3) for k, v in stage_ms.items()},
            "stage_ms_isolated": {k: round(v, 3) for k, v in iso_ms.items()},
        }
        # ---------------- CPU baseline + roofline of the dominant kernel (N = 1 only)
        if world == 1:
            ncores = os.cpu_count() or 1
            cb = cpu_baseline(g, idx, reads, min(a.cpu_sample, n_reads), ncores)
            t_cpu = cb["t_seed"] + cb["t_ext"] + (0.0 if a.host_jobs else cb["t_chain"])
            cpu_mreads = cb["n"] / t_cpu / 1e6
            res["cpu_baseline"] = {"value": round(cpu_mreads, 5), "unit": "Mreads/s", "cores": ncores, "kind": "port",
                                   "sample": f"first {cb['n']} reads of the same batch: oracle seeding {cb['t_seed']:.2f}s + "
                                             + ("" if a.host_jobs else f"host chaining/job builder (bmh_build_jobs) {cb['t_chain']:.2f}s + ")
                                             + f"oracle extension {cb['t_ext']:.2f}s on {ncores} threads"}
            res["speedup_vs_cpu_baseline"] = round(value / cpu_mreads, 1)
            res["oracle_work_per_read"] = {k: round(v / cb["n"], 2) for k, v in cb["work"].items()}
            # algorithmic bytes per read, counted by the oracle on the sample
            wk, n = cb["work"], cb["n"]
            fused = "smem" in stage_ms
            per_read = {"locate": (32.0 * wk["n_blk_lf"] + 4.0 * wk["n_sa"] + 20.0 * cb["n_seeds"]) / n}
            if fused:
                per_read["smem"] = 32.0 * (wk["n_blk_fwd"] + wk["n_blk_back"]) / n + a.read_len / 4 + 8
            else:
                per_read["forward"] = 32.0 * wk["n_blk_fwd"] / n + a.read_len / 4 + 8
                per_read["backward"] = 32.0 * wk["n_blk_back"] / n
            q, t = (jobs.qlen.long(), jobs.tlen.long()) if a.host_jobs else (jq.long(), jt.long())
            ext_bytes = float(((q + 3) // 4 + (t + 3) // 4 + (q + t + 7) // 8 + 28).sum().item())
            kernel_bytes = {k: v * n_reads for k, v in per_read.items()}
            kernel_bytes["extend"] = ext_bytes
            # Dominant kernel = the one rocprofv3 --stats ranks first among this library's kernels.  In round 1 that is
            # the largest class of the extension family (extend16_kernel<8>), which is integer-VALU bound: its HBM
            # fraction is ~0 by nature, so the same object also carries the stage's cell rate, and the
            # dominant HBM-bound kernel (the SMEM backward search) is reported beside it.
            def hbm_obj(k, name):
                ach_ = kernel_bytes[k] / (iso_ms[k] * 1e-3) / 1e9
                return {"bound": "hbm", "kernel": name, "achieved": round(ach_, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ach_ / HBM_PEAK_GBS, 5), "traffic": pmc_traffic(name), "avg_ms": round(iso_ms[k], 3),
                        "algorithmic_bytes_per_launch": int(kernel_bytes[k])}
            names = {"smem": "smem_fused_kernel", "forward": "smem_forward_kernel", "backward": "smem_backward_kernel", "locate": "locate_kernel",
                     "extend": "extend16_kernel<1..18> (class kernels on concurrent streams)"}
            dom = max(kernel_bytes.keys(), key=lambda k: iso_ms.get(k, 0.0))
            res["roofline"] = hbm_obj(dom, names[dom])
            if dom == "extend":
                res["roofline"]["note"] = ("integer-VALU bound DP (no MFMA, ~110 B of HBM traffic per job; SQ counters: VALU busy "
                                           "fraction in extension_stage.valu_busy_frac): see extension_stage for its cell "
                                           "rate and roofline_hbm_kernel for the dominant HBM-bound kernel")
            hb = max((k for k in kernel_bytes if k != "extend"), key=lambda k: iso_ms.get(k, 0.0))
            res["roofline_hbm_kernel"] = hbm_obj(hb, names[hb])
            res["roofline_hbm_kernel"]["gather_ceiling_GBps"] = GATHER_CEILING_GBS
            res["roofline_hbm_kernel"]["valu_busy_frac"] = pmc_valu_busy(lambda k: names[hb].split("<")[0] in k)
            res["roofline_hbm_kernel"]["note"] = ("random 32-byte index-block gathers; measured chip ceiling for this pattern = 56.8 G gathers/s = "
                                                  "1818 GB/s of useful bytes (scripts/calib.py); algorithmic bytes count every block the CPU "
                                                  "algorithm touches, cache hits included")
            res["roofline_all"] = {k: {"ms": round(iso_ms[k], 3), "algorithmic_GBps": round(kernel_bytes[k] / (iso_ms[k] * 1e-3) / 1e9, 2)}
                                   for k in kernel_bytes}
            cells = cb["cells"] / cb["n_jobs"] * n_jobs
            res["extension_stage"] = {"bound": "integer VALU (not HBM, not MFMA)", "ms": round(iso_ms["extend"], 3),
                                      "valu_busy_frac": pmc_valu_busy(lambda k: "extend16_kernel" in k or "extend_wide_kernel" in k),
                                      "gcups_reference_cells": round(cells / (iso_ms["extend"] * 1e-3) / 1e9, 1),
                                      "jobs": n_jobs, "hbm_GBps": round(kernel_bytes["extend"] / (iso_ms["extend"] * 1e-3) / 1e9, 2)}
            if not a.host_jobs:
                try:
                    res["next_rows"] = downstream_stages(L, dindex, dr, cw, regs_out, n_regs, n_reads, g, pac_t, reads, params, a.paired)
                except Exception as e:                      # never lose the bench line over the extras
                    res["next_rows"] = {"error": repr(e)}
        print(json.dumps(res), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
