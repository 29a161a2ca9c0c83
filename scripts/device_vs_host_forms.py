"""One-off parity run at bench scale: bmh_aligner_run on N reads against the hg38-scale synthetic index, the device forms (selection, packed CIGARs, SAM text,
pairing on the device) against the host forms (BMH_ALIGNER_HOST_FORMAT, BMH_ALIGNER_PE_HOST) and, paired, with the rescue's windows found on the device
(BMH_ALIGNER_RESCUE_DEV, the host's first walk beside it: BMH_RESCUE_CHECK): sha256 of the text per mode.  FORMS_ONLY=pe|se: one of the two halves.
usage: device_vs_host_forms.py [genome_mbp] [n_reads] [read_len]      FORMS_ALT=<n>: the last n sequences are ALT contigs (single-end: the device tail's ALT rules
against the host tail's, BMH_ALIGNER_ALT_HOST_PATCH)"""
import os, sys, hashlib, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bwa-mem_gpu_amd"))
import numpy as np, torch
import bwamem_hip as B
from bwamem_hip import fmindex as F
from bwamem_hip.aligner import ReadSet
from bwamem_hip.lib import NativeAligner, PeOpt, ChainOpt, PostOpt
mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 3100
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
rl = int(sys.argv[3]) if len(sys.argv) > 3 else 150
dev = torch.device("cuda:0")
L = B.load_library()
n_genome = int(mbp * 1e6)
g_t, meta = B.synth.make_genome_device(n_genome, dev, seed=42, return_meta=True)
pac_t = F.pack_pac_device(g_t)
g = g_t.cpu().numpy(); del g_t
torch.cuda.empty_cache()
d = F.build_fmd_index_device(pac_t, n_genome, sa_intv=1)
dindex = B.Index.from_device(d.primary, d.L2.astype(np.uint64), d.seq_len, d.bwt_t, d.sa_intv, d.sa_t, d.bits_t, pac_t=pac_t, l_pac=n_genome)
torch.cuda.empty_cache()
co = ChainOpt(); L.bmh_chain_opt_default(C.byref(co)); po = PostOpt(); L.bmh_post_opt_default(C.byref(po)); pe_o = PeOpt(); L.bmh_pe_opt_default(C.byref(pe_o))
n_alt = int(os.environ.get("FORMS_ALT", "0"))
is_alt = None
if n_alt:
    is_alt = np.zeros(len(meta["contigs"]), np.uint8); is_alt[-n_alt:] = 1
    print("ALT contigs:", [c[0] for c in meta["contigs"][-n_alt:]], flush=True)
nat = NativeAligner(dindex, pac_t.cpu().numpy(), n_genome, meta["contigs"], is_alt, co, B.ExtParams.default(), po, pe_o)
nth = L.bmh_effective_cpus()
bad = 0
for paired in [q for q in (False, True) if os.environ.get("FORMS_ONLY", "") in ("", "pe" if q else "se")]:
    for seed in (101, 202):
        reads = (B.synth.make_pairs(g, n_reads // 2, rl, seed=seed, holes=meta["holes"]) if paired else B.synth.make_reads(g, n_reads, rl, seed=seed, holes=meta["holes"]))[0]
        flat = np.ascontiguousarray(np.asarray(reads, np.uint8).reshape(-1))
        w = len(str(n_reads))
        names = np.char.add("r", np.char.zfill((np.arange(n_reads) // (2 if paired else 1)).astype(str), w))
        blob = np.frombuffer(("\0".join(names.tolist()) + "\0").encode(), dtype=np.uint8)
        rs = ReadSet(B.synth.codes_to_ascii(flat), np.arange(n_reads, dtype=np.uint64) * np.uint64(rl), np.full(n_reads, rl, np.uint32), blob,
                     np.arange(n_reads, dtype=np.uint64) * np.uint64(w + 2), codes=flat)
        cuts = [0, (n_reads // 2) & ~1, n_reads]
        hs = {}
        for env in ("", "BMH_ALIGNER_HOST_FORMAT") + (("BMH_ALIGNER_PE_HOST", "BMH_ALIGNER_RESCUE_DEV") if paired else ()) + (("BMH_ALIGNER_ALT_HOST_PATCH",) if n_alt and not paired else ()):
            if env:
                os.environ[env] = "1"
            if env == "BMH_ALIGNER_RESCUE_DEV":
                os.environ["BMH_RESCUE_CHECK"] = "1"
            h = hashlib.sha256(); nb = [0]
            def sink(mv):
                h.update(mv); nb[0] += len(mv)
            nat.run(rs, cuts, paired, sink, n_lanes=2, n_threads=nth)
            if env:
                del os.environ[env]
            if os.environ.pop("BMH_RESCUE_CHECK", None):
                chk = (C.c_uint64 * 5)(); L.bmh_rescue_check_counts(chk)
                print("   rescue windows on the device vs the host's first walk, so far: batches %d, pairs %d, alignments asked for %d, pairs whose active flag differs %d, pairs whose call list differs %d" % tuple(chk), flush=True)
                bad += chk[3] != 0
            hs[env or "device"] = (h.hexdigest()[:16], nb[0])
        ok = len(set(hs.values())) == 1
        bad += not ok
        print("paired" if paired else "single", "seed", seed, n_reads, "reads of", rl, "bp:", hs, "IDENTICAL" if ok else "DIFFERENT", flush=True)
nat.free()
sys.exit(1 if bad else 0)
