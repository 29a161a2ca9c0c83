"""Native reads -> SAM pipeline (bmh_aligner_run) on the bench workload (hg38-scale synthetic index built on the device): lanes x batch
counts, with the per-batch timeline of the last run (BMH_ALIGNER_TRACE).  usage: lanes_probe.py [genome_mbp] [n_reads] [pe]"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bwa-mem_gpu_amd"))
import numpy as np
if os.environ.get("LANES_BLOCKING"):
    _hip = C.CDLL("libamdhip64.so")
    print("hipSetDeviceFlags(blocking sync) ->", _hip.hipSetDeviceFlags(C.c_uint(4)))
import torch
import bwamem_hip as B
from bwamem_hip import fmindex as F
from bwamem_hip.aligner import ReadSet
from bwamem_hip.lib import NativeAligner, PeOpt, ChainOpt, PostOpt
mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 3100
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
paired = len(sys.argv) > 3 and sys.argv[3] == "pe"
dev = torch.device("cuda:0")
L = B.load_library()
n_genome = int(mbp * 1e6)
g_t, meta = B.synth.make_genome_device(n_genome, dev, seed=42, return_meta=True)
pac_t = F.pack_pac_device(g_t)
g = g_t.cpu().numpy()
del g_t
torch.cuda.empty_cache()
d = F.build_fmd_index_device(pac_t, n_genome, sa_intv=int(os.environ.get("SA_INTV", "1")))
contigs, holes = meta["contigs"], meta["holes"]
dindex = B.Index.from_device(d.primary, d.L2.astype(np.uint64), d.seq_len, d.bwt_t, d.sa_intv, d.sa_t, d.bits_t, pac_t=pac_t, l_pac=n_genome)
torch.cuda.empty_cache()
rl = int(os.environ.get("LANES_READ_LEN", "150"))
reads = (B.synth.make_pairs(g, n_reads // 2, rl, seed=7, holes=holes) if paired else B.synth.make_reads(g, n_reads, rl, seed=7, holes=holes))[0]
flat = np.ascontiguousarray(np.asarray(reads, np.uint8).reshape(-1))
offs = np.arange(n_reads, dtype=np.uint64) * np.uint64(rl)
co = ChainOpt(); L.bmh_chain_opt_default(C.byref(co)); po = PostOpt(); L.bmh_post_opt_default(C.byref(po))
params = B.ExtParams.default()
pe_o = PeOpt(); L.bmh_pe_opt_default(C.byref(pe_o))
pac_h = pac_t.cpu().numpy()
asc = B.synth.codes_to_ascii(flat)
w = len(str(n_reads))
names = np.char.add("r", np.char.zfill((np.arange(n_reads) // (2 if paired else 1)).astype(str), w))
blob = np.frombuffer(("\0".join(names.tolist()) + "\0").encode(), dtype=np.uint8)
noff = np.arange(n_reads, dtype=np.uint64) * np.uint64(w + 2)
rs = ReadSet(asc, offs, np.full(n_reads, rl, np.uint32), blob, noff, codes=flat)
n_alt = int(os.environ.get("LANES_ALT", "0"))                 # the last LANES_ALT sequences flagged as ALT contigs
is_alt = None
if n_alt:
    is_alt = np.zeros(len(contigs), np.uint8); is_alt[-n_alt:] = 1
    print("ALT contigs:", [c[0] for c in contigs[-n_alt:]], "%.1f %% of the genome" % (100.0 * sum(c[1] for c in contigs[-n_alt:]) / len(g)), flush=True)
nat = NativeAligner(dindex, pac_h, len(g), contigs, is_alt, co, params, po, pe_o)
L.bmh_host_pin.argtypes = [C.c_void_p, C.c_size_t]
pinned = os.environ.get("LANES_PIN", "1") != "0" and L.bmh_host_pin(asc.ctypes.data, asc.nbytes) == 0       # LANES_PIN=0: pageable letters, staged by the lanes' host threads
nbytes = [0]
def sink(mv): nbytes[0] += len(mv)
nth = int(os.environ.get("LANES_THREADS", "0")) or L.bmh_effective_cpus()
def throttled():
    try:
        d = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
        return int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", 0)) / 1e3, int(d.get("usage_usec", 0)) / 1e3
    except Exception:
        return 0, 0.0, 0.0
cfgs = [tuple(int(v) for v in c.split("x")) for c in os.environ.get("LANES_CFGS", "2x4,3x4,3x8,4x8,2x8,2x4").split(",")]
cfgs = [(l_, b_, int(s_)) for s_ in os.environ.get("LANES_SLOTS", "-1").split(",") for l_, b_ in cfgs]      # LANES_SLOTS: values of the knob ALIGNER_GPU_SLOTS (-1: leave it alone)
# LANES_KNOB=NAME:v1,v2[:rounds]: every configuration under these values of the knob NAME in turn, `rounds` times over (an A/B on one box)
if os.environ.get("LANES_KNOB"):
    kn = os.environ["LANES_KNOB"].split(":")
    cfgs = [(l_, b_, s_, (kn[0], int(v))) for _ in range(int(kn[2]) if len(kn) > 2 else 1) for l_, b_, s_ in cfgs for v in kn[1].split(",")]
else:
    cfgs = [c + (None,) for c in cfgs]
for lanes, nb, slots, knob in cfgs:
    if slots >= 0:
        L.bmh_tune_set(b"ALIGNER_GPU_SLOTS", slots, 0); print("ALIGNER_GPU_SLOTS = %d" % slots)
    if knob:
        L.bmh_tune_set(knob[0].encode(), knob[1], 0); print("%s = %d" % knob)
    q = (n_reads // nb) & ~1
    cuts = [k * q for k in range(nb)] + [n_reads]
    n_it = int(os.environ.get("LANES_ITERS", "3"))
    all_ms = []
    for it in range(n_it):
        if it == n_it - 1 and os.environ.get("LANES_TRACE"): os.environ["BMH_ALIGNER_TRACE"] = "1"
        nbytes[0] = 0
        th0 = throttled()
        st = nat.run(rs, cuts, paired, sink, n_lanes=lanes, n_threads=nth)
        th1 = throttled()
        all_ms.append(round(st.seconds * 1e3, 1))
        os.environ.pop("BMH_ALIGNER_TRACE", None)
    print("   [cgroup] throttled %d times for %.1f ms during the run; CPU time used %.0f ms (%d threads asked for)" % (th1[0] - th0[0], th1[1] - th0[1], th1[2] - th0[2], nth))
    if n_it > 3: print("   every run, ms:", all_ms, "best %.2f Mreads/s, median %.2f" % (n_reads / min(all_ms[1:]) / 1e3, n_reads / sorted(all_ms[1:])[len(all_ms[1:]) // 2] / 1e3))
    print("lanes %d batches %d: %.1f ms = %.2f Mreads/s (%d bytes); format %.1f; lanes summed: H2D %.1f seed %.1f cem %.1f tail %.1f select %.1f cigar %.1f" %
          (lanes, nb, st.seconds * 1e3, n_reads / st.seconds / 1e6, nbytes[0], st.format_seconds * 1e3, st.h2d_seconds * 1e3, st.seed_seconds * 1e3,
           st.chain_extend_seconds * 1e3, st.tail_seconds * 1e3, st.select_seconds * 1e3, st.cigar_seconds * 1e3), flush=True)
    print("   copies (events on the lanes' streams): H2D %.1f MB in %.2f ms = %.1f GB/s, text D2H %.1f MB in %.2f ms = %.1f GB/s; reads %s; waits for a device slot %.1f ms" %
          (st.h2d_bytes / 1e6, st.h2d_copy_seconds * 1e3, st.h2d_bytes / max(st.h2d_copy_seconds, 1e-9) / 1e9, st.d2h_bytes / 1e6, st.d2h_copy_seconds * 1e3,
           st.d2h_bytes / max(st.d2h_copy_seconds, 1e-9) / 1e9, "in registered host memory" if pinned else "pageable (staged by the lanes)", st.gate_wait_seconds * 1e3), flush=True)
if os.environ.get("LANES_FILE"):
    # the same reads from a FILE: bmh_aligner_run_fasta (a loader thread ahead of the lanes) against loading the file first (bmh_reads_load_fasta) and bmh_aligner_run
    from bwamem_hip.lib import load_fasta_reads
    fa = "/tmp/lanes_probe.fa"
    recs = np.empty((n_reads, w + 3 + rl + 1), np.uint8)
    recs[:, 0] = ord(">"); recs[:, 1:w + 2] = np.frombuffer("".join(names.tolist()).encode(), np.uint8).reshape(n_reads, w + 1); recs[:, w + 2] = 10
    recs[:, w + 3:w + 3 + rl] = asc.reshape(n_reads, rl); recs[:, -1] = 10
    recs.tofile(fa); del recs
    for lanes, nb, _, _ in cfgs:
        for it in range(3):
            nbytes[0] = 0
            t0 = time.perf_counter()
            st = nat.run_fasta(fa, paired, sink, batch_reads=(n_reads // nb) & ~1, n_lanes=lanes, n_threads=nth)
            dt = time.perf_counter() - t0
        print("FILE  lanes %d batches %d: file -> SAM %.1f ms = %.2f Mreads/s (%d bytes)" % (lanes, nb, dt * 1e3, n_reads / dt / 1e6, nbytes[0]), flush=True)
    t0 = time.perf_counter(); d_ = load_fasta_reads(fa); t_load = time.perf_counter() - t0
    print("FILE  bmh_reads_load_fasta alone: %.1f ms (%d reads)" % (t_load * 1e3, len(d_["lens"])), flush=True)
nat.free()
