"""End-to-end drop-in run: the REFERENCE's own `bwa-gasal2 gase_aln` host code (compiled unchanged by
scripts/build_dropin.sh, linked against libbwamem_hip.so) on BASELINE.json configs[0]'s shape:
10k synthetic 150 bp single-end reads vs an E. coli-size (4.64 Mbp) seeded genome.  Checks the SAM
against the simulation truth (position/strand of every read).
-K keeps the whole file in one batch whatever -t is (insert-size statistics are per batch).  NOTE: with -t >= 4 the
reference's own host code is not deterministic on hard read sets (regions with garbage scores, e.g. AS:i:7227, that
change from run to run): mem_align1_core indexes seq[] with a batch-relative read index where it collects its results
(src/bwamem.c:2228, 2251, 2295, 2313), so the patch test aligns another worker's read, possibly while that worker converts
it in place (scripts/e2e_seqidx_probe.sh; E2E_EXE=bwa-gasal2-seqidx runs the build with those four
subscripts corrected).  The stock build is compared at -t 1 with -K, where relative and absolute index coincide."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bwa-mem_gpu_amd"))
import numpy as np
import torch
from bwamem_hip import fmindex, synth

exe = os.path.join(ROOT, "build", "dropin", os.environ.get("E2E_EXE", "bwa-gasal2"))      # E2E_EXE=bwa-gasal2-seqidx: the build with the reference's four batch-relative seq[] indices corrected (scripts/build_dropin.sh)
if not os.path.exists(exe):
    sys.exit("build/dropin/bwa-gasal2 missing: run scripts/build_dropin.sh in the build container")
work = sys.argv[1] if len(sys.argv) > 1 else "/tmp/e2e_dropin"
n_genome = int(float(sys.argv[2])) if len(sys.argv) > 2 else 4_640_000
n_reads = int(float(sys.argv[3])) if len(sys.argv) > 3 else 10_000
threads = sys.argv[4] if len(sys.argv) > 4 else "1"
paired = len(sys.argv) > 5 and sys.argv[5] in ("pe", "pe_hard")
hard = len(sys.argv) > 5 and sys.argv[5] in ("se_hard", "pe_hard")      # diverged / chimeric / unalignable reads mixed in
opts = sys.argv[6].split() if len(sys.argv) > 6 else []               # gase_aln options given to both sides, e.g. "-k 23 -A 2 -B 5 -a"
os.makedirs(work, exist_ok=True)
prefix = os.path.join(work, "g.fa")
import ast
g = synth.make_genome(n_genome, seed=42, **ast.literal_eval(os.environ.get("E2E_GENOME_KW", "{}")))     # e.g. "{'repeat_frac': 0.6, 'repeat_copies': (2000, 5000)}"
t = time.time()
if os.environ.get("E2E_NATIVE_BUILD"):       # genomes beyond the torch builder's reach (seq_len > 2^32: E2E_NATIVE_BUILD=1 with a genome of 2.2e9 bases): bmh_index_build on the device, the files from its arrays
    g_t = torch.from_numpy(g).cuda()
    pac_t = fmindex.pack_pac_device(g_t); del g_t
    d = fmindex.build_fmd_index_device(pac_t, n_genome, sa_intv=16)
    idx = fmindex.FMDIndex(primary=int(d.primary), L2=np.asarray(d.L2, dtype=np.int64), seq_len=int(d.seq_len), bwt_words=d.bwt_t.cpu().numpy().view(np.uint32),
                           sa_intv=16, n_sa=int(d.sa_t.numel()), sa=d.sa_t.cpu().numpy().view(np.uint32), sa_bits=d.bits_t.cpu().numpy().view(np.uint32), pack_size=1)
    del d, pac_t; torch.cuda.empty_cache()
else:
    idx = fmindex.build_fmd_index(g, device="cuda:0" if torch.cuda.is_available() else None)
# E2E_CONTIGS=k: the genome is written as k sequences of unequal lengths (reads that straddle a cut lose the seeds that bridge it
# and get their extension windows clipped, src/bwamem.c:437, src/bntseq.c:531-556; positions are reported per sequence)
def rescue_check_line():
    """with BMH_RESCUE_CHECK (and BMH_ALIGNER_RESCUE_DEV=1): what the host's walk beside the device's search for the rescue's windows found"""
    if os.environ.get("BMH_RESCUE_CHECK"):
        import ctypes as _C
        from bwamem_hip.lib import load_library as _ll
        _chk = (_C.c_uint64 * 5)(); _ll().bmh_rescue_check_counts(_chk)
        print("   rescue windows on the device vs the host's walk: batches %d, pairs %d, alignments asked for %d, pairs whose active flag is differing %d, pairs whose call list is differing %d" % tuple(_chk))
n_ctg = int(os.environ.get("E2E_CONTIGS", "1"))
contigs = None
if n_ctg > 1:
    cuts = sorted(set(int(x) for x in np.random.default_rng(5).integers(1000, n_genome - 1000, size=n_ctg - 1)))
    edges = [0] + cuts + [n_genome]
    contigs = [("ctg%d" % i, edges[i + 1] - edges[i]) for i in range(len(edges) - 1)]
fmindex.write_index(prefix, idx); fmindex.write_bns(prefix, g, contigs=contigs)
# E2E_ALT=k (with E2E_CONTIGS): the last k sequences are ALT contigs, named in <prefix>.alt the way bwa-kit's files name them (both sides read the file)
if os.path.exists(prefix + ".alt"):
    os.remove(prefix + ".alt")
if contigs and int(os.environ.get("E2E_ALT", "0")) > 0:
    with open(prefix + ".alt", "w") as f:
        f.write("@SQ\tSN:%s\n" % contigs[-1][0])
        for c in contigs[-int(os.environ["E2E_ALT"]):]:
            f.write("%s\t0\t%s\t1\t60\t100M\t*\t0\t0\t*\t*\n" % (c[0], contigs[0][0]))
print("index built+written in %.1fs" % (time.time() - t), flush=True)
fq = os.path.join(work, "reads.fa")
if paired:   # configs[3]: one interleaved file with -p (the only coherent PE input of the reference, SURVEY.md 8 notes)
    RL = int(os.environ.get("E2E_READLEN", "150"))
    reads, ptruth = synth.make_pairs(g, n_reads // 2, RL, seed=7, sub_rate=0.03 if hard else 0.01)
    if hard:
        rng = np.random.default_rng(8); L = RL
        for i in range(0, len(reads), 2):
            kind = (i // 2) % 10
            m = i + int(rng.integers(0, 2))
            if kind == 3:
                x = reads[m]; q = rng.random(L) < 0.12; x[q] = (x[q] + rng.integers(1, 4, size=int(q.sum()))) & 3
            elif kind == 5:
                p0 = int(rng.integers(0, n_genome - L)); x = g[p0:p0 + L].copy(); reads[m] = x if rng.random() < 0.5 else synth.revcomp(x)
            elif kind == 7:
                reads[m] = rng.integers(0, 4, size=L).astype(np.uint8)
            elif kind == 9:
                k = int(rng.integers(50, 100)); p1 = int(rng.integers(0, n_genome - L)); reads[m][k:] = g[p1:p1 + L - k]
    asc = synth.codes_to_ascii(reads)
    with open(fq, "wb") as f:
        for i in range(asc.shape[0]):
            # E2E_READNO=1: names "p<k>/1", "p<k>/2 comment" -- both sides cut the comment and the read number (trim_readno, src/bwa.c:27-31)
            f.write((b">p%d/%d%s\n" % (i // 2, i % 2 + 1, b" x:y z" if i % 3 == 0 else b"")) if os.environ.get("E2E_READNO") else (b">p%d\n" % (i // 2))); f.write(asc[i].tobytes()); f.write(b"\n")
else:
    RL = int(os.environ.get("E2E_READLEN", "150"))
    reads, truth = synth.make_reads(g, n_reads, RL, seed=7, sub_rate=0.04 if hard else 0.01, indel_frac=0.4 if hard else 0.05,
                                    n_rate=float(os.environ.get("E2E_NRATE", "0.001")))
    if hard:
        rng = np.random.default_rng(6); L = RL
        for i in range(3, n_reads, 12):                      # chimeric reads
            k = int(rng.integers(50, 100)); p1 = int(rng.integers(0, n_genome - L)); b = g[p1:p1 + L - k].copy()
            reads[i][k:] = synth.revcomp(b) if rng.random() < 0.5 else b
    if os.environ.get("E2E_LONGDEL"):         # every E2E_LONGDEL-th read spans a 12..60 bp deletion: two colinear regions -> mem_patch_reg's global alignment
        rng = np.random.default_rng(10)
        for i in range(1, n_reads, int(os.environ["E2E_LONGDEL"])):
            p0 = int(rng.integers(0, n_genome - RL - 80)); d = int(rng.integers(12, 60)); h = RL // 2
            x = np.concatenate([g[p0:p0 + h], g[p0 + h + d:p0 + RL + d]])
            reads[i] = x if rng.random() < 0.5 else synth.revcomp(x)
    if os.environ.get("E2E_RAGGED"):          # reads cut to lengths between E2E_RAGGED_MIN (20; the reference itself aborts on some very short reads: ks_resize of 0 bytes) and the full length, hard mode only
        assert hard
        asc = synth.codes_to_ascii(reads); lens_r = np.random.default_rng(9).integers(int(os.environ.get("E2E_RAGGED_MIN", "20")), RL + 1, size=n_reads)
        with open(fq, "wb") as f:
            for i in range(n_reads):
                f.write(b">r%d\n" % i); f.write(asc[i, :lens_r[i]].tobytes()); f.write(b"\n")
    elif os.environ.get("E2E_READNO"):
        asc = synth.codes_to_ascii(reads)
        with open(fq, "wb") as f:
            for i in range(n_reads):
                f.write(b">r%d/%d\tcomment\n" % (i, 1 + i % 2)); f.write(asc[i].tobytes()); f.write(b"\n")
    else:
        synth.write_fasta_reads(fq, reads)
sam = os.path.join(work, "out.sam")
t = time.time()
with open(sam, "w") as f:
    # E2E_DEFAULT_K=1: no -K -- the reference cuts its batches itself (10 Mbases per thread, src/fastmap.c:527; the insert-size statistics are per batch) and the aligner
    # below cuts the same way (align_file's default); only meaningful with E2E_EXE=bwa-gasal2-seqidx: the stock build indexes seq[] batch-relative (see the docstring)
    kopt = [] if os.environ.get("E2E_DEFAULT_K") else ["-K", "2000000000"]
    r = subprocess.run([exe, "gase_aln", "-t", threads] + kopt + ["-l", os.environ.get("E2E_READLEN", "150")] + opts + (["-p"] if paired else []) + [prefix, fq], stdout=f, stderr=subprocess.PIPE, cwd=work)
dt = time.time() - t
print("gase_aln rc=%d in %.2fs" % (r.returncode, dt))
print(r.stderr.decode()[-1500:])
if r.returncode != 0:
    sys.exit(1)
if paired:
    n = mapped = proper = okpos = 0
    for line in open(sam):
        if line[0] == "@":
            continue
        c = line.split("\t"); flag = int(c[1])
        if flag & 0x900:
            continue
        n += 1
        if not flag & 4:
            mapped += 1
        if flag & 2:
            proper += 1
        i = int(c[0][1:])
        pos = int(c[3]) - 1
        lo = int(ptruth["pos"][i]); hi = lo + int(ptruth["insert"][i])
        if not flag & 4 and lo - 8 <= pos <= hi + 8:
            okpos += 1
    print(f"PE reads {n} mapped {mapped} properly paired {proper} ({proper/max(n,1):.4f}) inside the simulated fragment {okpos} ({okpos/max(n,1):.4f})")
    assert n == 2 * (n_reads // 2) and (hard or (proper / n > 0.95 and okpos / n > 0.97))
    # the same job through the device-resident path: one batch like the reference's (its insert-size statistics are per batch)
    from bwamem_hip.aligner import Aligner
    import io
    al = Aligner(prefix); al.set_options(opts + (["-t", threads] if os.environ.get("E2E_DEFAULT_K") else []))       # (-t: the reference's batches hold 10 Mbases per thread)
    buf = io.StringIO()
    al.align_file(fq, buf, batch_reads=0 if os.environ.get("E2E_DEFAULT_K") else 1 << 30, paired=True)
    ours = [l for l in buf.getvalue().split("\n") if l and l[0] != "@"]
    theirs = [l.rstrip("\n") for l in open(sam) if l[0] != "@"]
    diff = [(a, b) for a, b in zip(ours, theirs) if a != b]
    print(f"device-resident path: {len(ours)} records; reference host code: {len(theirs)} records; differing records: {len(diff)}")
    rescue_check_line()
    if diff:                                              # keep the evidence: both records and the pair's reads
        os.makedirs("gpurun_out", exist_ok=True)
        with open("gpurun_out/e2e_diff_%s.txt" % os.environ.get("E2E_TAG", "pe_t" + threads), "w") as f:
            f.write(f"{len(diff)} differing records of {len(ours)}; options {opts}\n")
            for a, b in diff[:40]:
                i = int(a.split("\t")[0][1:])
                f.write("OURS   " + a + "\nTHEIRS " + b + "\n")
                f.write("R1 " + asc[2 * i].tobytes().decode() + "\nR2 " + asc[2 * i + 1].tobytes().decode() + "\n")
            names = []
            for a, b in diff:
                nm = a.split("\t")[0]
                if nm not in names: names.append(nm)
                if len(names) >= 8: break
            f.write("\nEVERY RECORD OF THE FIRST DIFFERING PAIRS (fields 1-9, tags)\n")
            for nm in names:
                for tag, lines in (("OURS  ", ours), ("THEIRS", theirs)):
                    for l in lines:
                        c = l.split("\t")
                        if c[0] == nm: f.write(tag + " " + "\t".join(c[:9]) + "\t" + "\t".join(c[11:]) + "\n")
    assert len(ours) == len(theirs) and not diff, diff[:2]
    print("SAM IDENTICAL")
    print("E2E DROP-IN OK")
    sys.exit(0)
ok = mapped = n = 0
flags = {}
for line in open(sam):
    if line[0] == "@":
        continue
    c = line.split("\t")
    flag = int(c[1])
    if flag & 0x900:
        continue
    n += 1
    i = int(c[0][1:])
    if flag & 4:
        continue
    mapped += 1
    pos = int(c[3]) - 1
    rev = bool(flag & 16)
    if abs(pos - int(truth["pos"][i])) <= 8 and rev == bool(truth["rev"][i]):
        ok += 1
print(f"reads {n} mapped {mapped} ({mapped/max(n,1):.4f}) correct position+strand {ok} ({ok/max(n,1):.4f})")
lines = [l for l in open(sam) if l[0] != "@"][:3]
print("".join(l[:200] + "\n" for l in lines))
assert n == n_reads and (hard or ok / n > 0.97), "end-to-end accuracy too low"
# the same job through the device-resident path (bwamem_hip.aligner): the records must equal the reference's, byte for byte
from bwamem_hip.aligner import Aligner
import io
t = time.time()
al = Aligner(prefix); al.set_options(opts)
buf = io.StringIO()
al.align_file(fq, buf, batch_reads=4096)                  # several batches: the tie-break hash depends on the global read index
dt2 = time.time() - t
ours = [l for l in buf.getvalue().split("\n") if l and l[0] != "@"]
theirs = [l.rstrip("\n") for l in open(sam) if l[0] != "@"]
diff = [(a, b) for a, b in zip(ours, theirs) if a != b]
print(f"device-resident path: {len(ours)} records in {dt2:.2f}s (reference host code: {len(theirs)} records in {dt:.2f}s); differing records: {len(diff)}")
if diff or len(ours) != len(theirs):
    os.makedirs("gpurun_out", exist_ok=True)
    tag = os.environ.get("E2E_TAG", "se")
    with open(f"gpurun_out/e2e_diff_{tag}.txt", "w") as f:
        f.write(f"{len(diff)} differing records of {len(ours)} / {len(theirs)}; options {opts}\n")
        import collections
        go, gt = collections.defaultdict(list), collections.defaultdict(list)
        for l in ours: go[l.split("\t")[0]].append(l)
        for l in theirs: gt[l.split("\t")[0]].append(l)
        bad = [k for k in go if go[k] != gt.get(k)][:15]          # reads whose record lists differ (a missing record shifts every later line)
        f.write(f"reads with differing records: {sum(1 for k in go if go[k] != gt.get(k))}\n")
        for tg, S in (("OURS  ", ours), ("THEIRS", theirs)):
            for l in S:
                if l.split("\t")[0] in bad:
                    f.write(tg + " " + l + "\n")
assert len(ours) == len(theirs) and not diff, diff[:2]
print("SAM IDENTICAL")
print("E2E DROP-IN OK")
