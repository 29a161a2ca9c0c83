#!/usr/bin/env python3
"""bmh_index_build on the GPU: equality with the torch prefix-doubling builder at small sizes (several chunk capacities, so the
bucket passes and the group-aligned chunks of the doubling rounds are exercised), then timing + complete verification at
hg38-like sizes.  usage: index_build_probe.py [mbp ...]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bwa-mem_gpu_amd"))
import bwamem_hip as B
from bwamem_hip import fmindex as F, synth

dev = torch.device("cuda", 0)
B.load_library()


def check(name, g, caps=(None, 12, 9)):
    ref = F.build_fmd_index(g, sa_intv=16, device="cuda:0")
    pac = F.pack_pac_device(torch.from_numpy(g).to(dev))
    for cap in caps:
        if cap is None: os.environ.pop("BMH_BUILD_CAP_LOG2", None)
        else: os.environ["BMH_BUILD_CAP_LOG2"] = str(cap)
        for intv in (16, 1):
            try:
                d = F.build_fmd_index_device(pac, len(g), sa_intv=intv, verify=True)
            except RuntimeError as e:
                print(f"  {name} cap={cap} intv={intv}: {e}")
                continue
            h = F.device_index_to_host(d, 16)
            ok = (h.primary == ref.primary and np.array_equal(h.L2, ref.L2) and np.array_equal(h.bwt_words, ref.bwt_words)
                  and np.array_equal(h.sa, ref.sa) and np.array_equal(h.sa_bits, ref.sa_bits))
            print(f"  {name} n={2*len(g)} cap={cap} intv={intv}: {'OK' if ok else 'MISMATCH'} rounds={d.stats['doubling_rounds']} passes={d.stats['round0_passes']}")
            assert ok
    os.environ.pop("BMH_BUILD_CAP_LOG2", None)


if len(sys.argv) == 1:
    rng = np.random.default_rng(1)
    check("tiny", rng.integers(0, 4, 37, dtype=np.uint8))
    check("g20011", synth.make_genome(20011, seed=3))
    check("rep100k", synth.make_genome(100_000, seed=5, repeat_frac=0.5, repeat_div=0.01))
    check("polyA", np.zeros(3000, np.uint8), caps=(None, 14))
    check("tandem", np.tile(rng.integers(0, 4, 171, dtype=np.uint8), 40), caps=(None, 14))
    check("g1M", synth.make_genome(1_000_000, seed=7, repeat_frac=0.3, repeat_div=0.02), caps=(None, 16))
for mbp in [float(x) for x in sys.argv[1:]]:
    n = int(mbp * 1e6)
    gen = torch.Generator(device=dev); gen.manual_seed(42)
    t0 = time.time()
    g = synth.make_genome_device(n, dev, seed=42) if hasattr(synth, "make_genome_device") else torch.randint(0, 4, (n,), dtype=torch.uint8, device=dev, generator=gen)
    pac = F.pack_pac_device(g); del g
    torch.cuda.synchronize(); t_gen = time.time() - t0
    os.environ["BMH_BUILD_VERBOSE"] = "1"
    t0 = time.time()
    d = F.build_fmd_index_device(pac, n, sa_intv=1, verify=True)
    print(f"{mbp:g} Mbp: genome {t_gen:.1f}s, index build {time.time()-t0:.1f}s, stats {d.stats}, peak torch mem {torch.cuda.max_memory_allocated()/1e9:.1f} GB", flush=True)
    del d; torch.cuda.empty_cache()
