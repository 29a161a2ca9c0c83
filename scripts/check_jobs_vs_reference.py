"""Parity of the host job builder (bmh_build_jobs) with the REFERENCE's own host code.

Runs build/dropin/bwa-gasal2 (the reference's src/*.c, unchanged, on our library) with BMH_GASAL_DUMP
so every extension job its mem_chain/mem_chain_flt/mem_chain2aln submit is recorded, then builds the
jobs for the same reads with bmh_build_jobs from our seeds and compares the two job multisets
(h0, query bases, target bases) exactly.  Also checks the best region score of every read against
the AS tag of the reference's SAM output."""
import collections, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bwa-mem_gpu_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import bwamem_hip as B
from bwamem_hip import fmindex, synth
from bwamem_hip.lib import HostJobs

exe = os.path.join(ROOT, "build", "dropin", "bwa-gasal2")
work = sys.argv[1] if len(sys.argv) > 1 else "/tmp/jobs_vs_ref"
n_genome = int(float(sys.argv[2])) if len(sys.argv) > 2 else 2_000_000
n_reads = int(float(sys.argv[3])) if len(sys.argv) > 3 else 5000
L = int(sys.argv[4]) if len(sys.argv) > 4 else 150
extra = sys.argv[5:]                     # further gase_aln options, passed to the reference and mirrored in bmh_chain_opt_t: -W <min_chain_weight>
os.makedirs(work, exist_ok=True)
prefix = os.path.join(work, "g.fa")
g = synth.make_genome(n_genome, seed=42)
idx = fmindex.build_fmd_index(g, device="cuda:0" if torch.cuda.is_available() else None)
fmindex.write_index(prefix, idx); fmindex.write_bns(prefix, g)
reads, truth = synth.make_reads(g, n_reads, L, seed=7, sub_rate=0.02, indel_frac=0.15)
fq = os.path.join(work, "reads.fa"); synth.write_fasta_reads(fq, reads)
dump = os.path.join(work, "jobs.bin")
if os.path.exists(dump):
    os.remove(dump)
sam = os.path.join(work, "out.sam")
with open(sam, "w") as f:
    r = subprocess.run([exe, "gase_aln", "-t", "1", "-l", str(L)] + extra + [prefix, fq], stdout=f, stderr=subprocess.PIPE, cwd=work,
                       env=dict(os.environ, BMH_GASAL_DUMP=dump))
assert r.returncode == 0, r.stderr.decode()[-2000:]
raw = np.fromfile(dump, dtype=np.uint8)
ref_jobs = collections.Counter()
p = 0
while p < raw.size:
    ql, tl, h0 = np.frombuffer(raw[p:p + 12].tobytes(), dtype="<u4")
    p += 12
    ref_jobs[(int(h0), raw[p:p + ql].tobytes(), raw[p + ql:p + ql + tl].tobytes())] += 1
    p += int(ql) + int(tl)
# ours: GPU seeds -> host job builder
seeds = B.seed_file(prefix, fq, 19)
flat = reads.reshape(-1); offs = np.arange(n_reads, dtype=np.uint64) * L; lens = np.full(n_reads, L, np.uint32)
t0 = time.time()
from bwamem_hip.lib import ChainOpt
import ctypes as C
co = ChainOpt(); B.load_library().bmh_chain_opt_default(C.byref(co))
if "-W" in extra:
    co.min_chain_weight = int(extra[extra.index("-W") + 1])
hj = HostJobs(g, flat, offs, lens, seeds, n_threads=8, opt=co)
print("bmh_build_jobs: %d jobs, %d regions for %d reads in %.2fs" % (hj.n_jobs, hj.n_regs, n_reads, time.time() - t0))
ours = collections.Counter()
for i in range(hj.n_jobs):
    ours[(int(hj.h0[i]), hj.q[hj.qoff[i]:hj.qoff[i] + hj.qlen[i]].tobytes(), hj.t[hj.toff[i]:hj.toff[i] + hj.tlen[i]].tobytes())] += 1
only_ref = ref_jobs - ours; only_ours = ours - ref_jobs
print("reference jobs %d, ours %d, only-reference %d, only-ours %d" % (sum(ref_jobs.values()), sum(ours.values()), sum(only_ref.values()), sum(only_ours.values())))
assert not only_ref and not only_ours, "job multisets differ"
# the DEVICE job builder (bmh_chain_batch; with -W small enough, its forms with the reference's seed filter) on the same reads: the same multiset
if L <= 700:
    from bwamem_hip.lib import ChainWorkspace, dev_jobs_to_host
    import test_gpu_parity as tp_
    dindex = B.Index.upload(idx, pac=tp_._pack_pac(g), l_pac=len(g))
    ws = B.SeedWorkspace(n_reads, int(flat.size), max_cands=int(flat.size), max_occ=1 << 22)
    r_t = torch.from_numpy(synth.codes_to_ascii(flat)).cuda()
    o_t = torch.from_numpy(offs.astype(np.int64)).to(torch.int32).cuda(); l_t = torch.from_numpy(lens.astype(np.int64)).to(torch.int32).cuda()
    s_d = ws.seed_batch(dindex, r_t, o_t, l_t, 19)
    cw = ChainWorkspace(n_reads, max(int(s_d.n_seeds), 1), opt=co)
    dj = cw.chain_batch(dindex, r_t, o_t, l_t, s_d)
    got = dev_jobs_to_host(dj, n_reads)
    dev = collections.Counter()
    for i in range(int(dj.n_jobs)):
        dev[(int(got["h0"][i]), got["q"][got["qoff"][i]:got["qoff"][i] + got["qlen"][i]].tobytes(), got["t"][got["toff"][i]:got["toff"][i] + got["tlen"][i]].tobytes())] += 1
    print("device builder jobs %d, only-reference %d, only-device %d" % (sum(dev.values()), sum((ref_jobs - dev).values()), sum((dev - ref_jobs).values())))
    assert dev == ref_jobs, "device job builder: job multiset differs from the reference's"
    cw.free(); ws.free(); dindex.free()
# extend on the GPU, merge, compare best score per read with the SAM AS tag
import importlib
tp = importlib.import_module("test_gpu_parity")
out3, _ = tp.gpu_extend(B, hj.jobs(), want_raw=False) if hj.n_jobs else (np.zeros((0, 3), np.int32), None)
regs = hj.merge(out3)
best = np.full(n_reads, -1, np.int64)
np.maximum.at(best, regs[:, 0], regs[:, 1])
as_tag = {}
for line in open(sam):
    if line[0] == "@": continue
    c = line.rstrip("\n").split("\t")
    if int(c[1]) & 0x900: continue
    for tag in c[11:]:
        if tag.startswith("AS:i:"): as_tag[int(c[0][1:])] = int(tag[5:])
bad = [(i, int(best[i]), as_tag[i]) for i in as_tag if best[i] != as_tag[i]]
print("reads with AS tag %d, best-region score == AS for %d" % (len(as_tag), len(as_tag) - len(bad)), "first mismatches", bad[:5])
assert len(bad) <= 0.002 * len(as_tag) + 1
print("JOBS VS REFERENCE OK")
