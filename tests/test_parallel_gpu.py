"""The N > 1 HIP path on the hardware at hand: two ranks (fresh processes, one device shared when the box has one GPU) receive the
device-built index by broadcast, shard the reads and run seeding -> chaining -> extension -> merge; their outputs concatenate to
the single-process result and equal the oracle's.  Also: `python bench.py --gpus 2` without a launcher starts its own ranks."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import common

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("paired", [0, 1])
def test_two_ranks_run_the_hip_path_on_their_shards(oracle, tmp_path, paired):
    import torch
    import bwamem_hip as B
    from bwamem_hip import fmindex as F, synth
    from bwamem_hip.lib import HostJobs
    from bwamem_hip.parallel import rebase_prefix, shard_range
    assert torch.cuda.is_available()
    n_genome, n_reads, rl, world = 600_000, 3001 - paired, 150, 2
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "parallel_gpu_worker.py"), str(tmp_path), str(n_genome), str(n_reads), str(rl), str(paired)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=900)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    sh = [np.load(os.path.join(str(tmp_path), f"rank{r}.npz")) for r in range(world)]
    # the oracle on the whole read set (the same genome and reads, drawn here once more)
    dev = torch.device("cuda", 0)
    g_t, meta = synth.make_genome_device(n_genome, dev, seed=21, return_meta=True)
    g = g_t.cpu().numpy()
    idx = F.build_fmd_index(g)
    reads = (synth.make_pairs(g, n_reads // 2, rl, seed=5, holes=meta["holes"]) if paired else synth.make_reads(g, n_reads, rl, seed=5, holes=meta["holes"]))[0]
    spans = [shard_range(reads.shape[0], r, world, multiple=2 if paired else 1) for r in range(world)]
    assert [(int(s["lo"]), int(s["hi"])) for s in sh] == spans and all(lo % (2 if paired else 1) == 0 for lo, _ in spans)
    flat, offs, lens = common.flat_reads(reads)
    want = oracle.seed_reads(oracle.fmd(idx), flat, offs, lens, 19, n_threads=4)
    for k in ("rbeg", "qbeg", "score", "n_ref_pos"):
        assert np.array_equal(np.concatenate([s[k] for s in sh]), want[k]), k
    assert np.array_equal(rebase_prefix([s["prefix"] for s in sh], [s["n_ref_pos"] for s in sh]), want["prefix"])
    hj = HostJobs(g, flat, offs, lens, want, n_threads=4, contigs=meta["contigs"])
    want3, _, _ = oracle.extend_batch(*hj.jobs(), n_threads=4)
    want_regs = hj.merge(want3)
    assert sum(int(s["n_jobs"]) for s in sh) == hj.n_jobs
    got = []
    for s, (lo, _) in zip(sh, spans):
        r = s["regs"].copy(); r[:, 0] += lo            # shard-local read index -> index in the whole set
        got.append(r)
    assert np.array_equal(np.concatenate(got), want_regs)
    hj.free()


def test_bench_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE: two fresh ranks, one JSON line with n_gpus 2, the timed batch
    of every rank verified against the oracle (small workload: 40 Mbp genome, 20 k reads per rank)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--genome-mbp", "40", "--reads-per-gpu", "20000",
                        "--verify-sample", "2000", "--no-next-rows"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1200)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["value"] > 0 and res["scaling"] == "weak"
    v = res["verified"]
    assert v["seeds_identical"] and v["regions_identical"] and v["ranks"] == 2 and v["reads"] == 4000
