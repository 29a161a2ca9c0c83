"""Golden vector for the host job builder, produced on the GPU box by the REFERENCE's own host code:
runs build/dropin/bwa-gasal2 with BMH_GASAL_DUMP on a small seeded read set and stores, next to the inputs of
bmh_build_jobs (genome seed/size, reads, seeds), the multiset of extension jobs the reference submitted
(as sorted SHA-1 digests of h0|query|target) and the AS tag of every read.  Output: gpurun_out/jobs_golden.npz,
to be committed as tests/golden/jobs_golden.npz."""
import collections, hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bwa-mem_gpu_amd"))
import numpy as np, torch
import bwamem_hip as B
from bwamem_hip import fmindex, synth
# usage: make_jobs_golden.py            -> jobs_golden.npz  (plain genome)
#        make_jobs_golden.py repeats    -> post_golden.npz  (repeat-rich genome: secondary / supplementary alignments, XS, low MAPQ)
#        make_jobs_golden.py contigs    -> contigs_golden.npz (the same repeat-rich genome cut into three sequences; reads cross the cuts)
#        make_jobs_golden.py pe         -> pe_golden.npz (interleaved pairs, run with -p: insert-size statistics, mate rescue, pairing)
MODE = sys.argv[1] if len(sys.argv) > 1 else "plain"
#        make_jobs_golden.py pe_contigs -> pe_contigs_golden.npz (pairs on the three-sequence genome; some pairs span two sequences)
#        make_jobs_golden.py alt        -> alt_golden.npz (two primary sequences + two ALT contigs -- diverged copies of stretches of them -- named in g.fa.alt)
#        make_jobs_golden.py pe_alt     -> pe_alt_golden.npz (the same genome, interleaved pairs)
ALT = MODE in ("alt", "pe_alt", "pe_alt2")
REPEATS = MODE in ("repeats", "contigs", "pe", "pe_contigs", "alt", "pe_alt", "pe_alt2")
PE = MODE in ("pe", "pe_contigs", "pe_alt", "pe_alt2")
ALT2 = MODE == "pe_alt2"     # an ALT contig with a stretch the primary assembly does not have, and pairs whose mates reach it with nothing but a weak hit on the primary assembly
OUT = {"plain": "jobs_golden.npz", "repeats": "post_golden.npz", "contigs": "contigs_golden.npz", "pe": "pe_golden.npz", "pe_contigs": "pe_contigs_golden.npz",
       "alt": "alt_golden.npz", "pe_alt": "pe_alt_golden.npz", "pe_alt2": "pe_alt2_golden.npz"}[MODE]
CONTIGS = [("ctgA", 120_000), ("ctgB", 100_037), ("ctgC", 79_963)] if MODE in ("contigs", "pe_contigs") else None
if ALT:
    CONTIGS = [("ctgA", 150_000), ("ctgB", 110_000), ("altA1", 20_000), ("altB1", 12_000), ("altA2", 8_000)]
ALT_SRC = [(30_000, 20_000, 0.02), (170_000, 12_000, 0.04), (100_000, 8_000, 0.005)]      # (start on the primary assembly, length, divergence) of every ALT contig
work = "/tmp/jobs_golden_" + MODE; os.makedirs(work, exist_ok=True)
n_genome, n_reads, L = 300_000, 600, 150
GENOME_KW = dict(repeat_frac=0.45, repeat_len=(150, 1500), repeat_copies=(3, 40), repeat_div=0.03) if REPEATS else {}
g = synth.make_genome(n_genome, seed=42, **GENOME_KW)
if ALT:
    # the last 40 000 bases become ALT contigs: copies of stretches of the primary assembly with substitutions and a few short indels
    rng = np.random.default_rng(77)
    parts = [g[:260_000]]
    for (p0, ln, div) in ALT_SRC:
        x = g[p0:p0 + ln + 40].copy()
        m = rng.random(x.size) < div; x[m] = (x[m] + rng.integers(1, 4, size=int(m.sum()))) & 3
        for _k in range(3):                               # three indels of 1..6 bases
            q = int(rng.integers(500, ln - 500)); d = int(rng.integers(1, 7))
            x = np.concatenate([x[:q], x[q + d:]]) if rng.random() < 0.5 else np.concatenate([x[:q], rng.integers(0, 4, size=d).astype(np.uint8), x[q:]])
        parts.append(x[:ln])
    g = np.concatenate(parts).astype(np.uint8)
    assert g.size == n_genome
    if ALT2:                                              # 3 000 novel bases inside altA1 (an insertion the primary assembly lacks)
        g[265_000:268_000] = np.random.default_rng(78).integers(0, 4, size=3000).astype(np.uint8)
idx = fmindex.build_fmd_index(g, device="cuda:0")
prefix = os.path.join(work, "g.fa"); fmindex.write_index(prefix, idx); fmindex.write_bns(prefix, g, contigs=CONTIGS)
if ALT:
    with open(prefix + ".alt", "w") as f:           # SAM-like lines as bwa-kit ships them: header lines, then one line per ALT contig (first field = its name)
        f.write("@SQ\tSN:altA1\tLN:20000\n")
        f.write("altA1\t0\tctgA\t30001\t60\t20000M\t*\t0\t0\t*\t*\n")
        f.write("altB1\t16\tctgB\t20001\t60\t12000M\t*\t0\t0\t*\t*\n")
        f.write("not_in_the_index\t0\tctgA\t1\t60\t10M\t*\t0\t0\t*\t*\n")
        f.write("altA2\t0\tctgA\t100001\t60\t8000M\t*\t0\t0\t*\t*\r\n")
reads, _ = synth.make_reads(g, n_reads, L, seed=21, sub_rate=0.02, indel_frac=0.2)
if PE:
    reads, _ = synth.make_pairs(g, n_reads // 2, L, seed=21, sub_rate=0.02)
    rng = np.random.default_rng(8)
    for i in range(0, n_reads, 2):
        kind = (i // 2) % 10
        m = i + int(rng.integers(0, 2))                    # one mate of the pair
        if kind == 3:                                      # a mate too diverged to seed: found only by mate rescue
            x = reads[m]; q = rng.random(L) < 0.12; x[q] = (x[q] + rng.integers(1, 4, size=int(q.sum()))) & 3
        elif kind == 5:                                    # discordant: the mate comes from elsewhere
            p0 = int(rng.integers(0, n_genome - L)); x = g[p0:p0 + L].copy(); reads[m] = x if rng.random() < 0.5 else synth.revcomp(x)
        elif kind == 7:                                    # unalignable mate
            reads[m] = rng.integers(0, 4, size=L).astype(np.uint8)
if ALT:
    # half of the reads (or of the pairs' first mates) come from the stretches that have an ALT copy, or from the ALT contigs themselves
    rng = np.random.default_rng(12)
    offs = np.concatenate([[0], np.cumsum([c[1] for c in CONTIGS])])
    step = 2 if PE else 1
    for i in range(0, n_reads, 2 * step):
        k = int(rng.integers(0, len(ALT_SRC))); p0, ln, _ = ALT_SRC[k]
        on_alt = rng.random() < 0.5
        base = int(offs[2 + k]) if on_alt else p0
        if PE:
            ins = int(rng.integers(250, 450)); q = base + int(rng.integers(0, ln - ins))
            a = g[q:q + L].copy(); b = synth.revcomp(g[q + ins - L:q + ins])
            if rng.random() < 0.5: a, b = b, a
            reads[i], reads[i + 1] = a, b
        else:
            q = base + int(rng.integers(0, ln - L)); x = g[q:q + L].copy()
            m = rng.random(L) < 0.01; x[m] = (x[m] + rng.integers(1, 4, size=int(m.sum()))) & 3
            reads[i] = x if rng.random() < 0.5 else synth.revcomp(x)
if ALT2:
    # pairs that reach mem_sam_pe's rules for a read whose best hit on the primary assembly is below the threshold while it has a good ALT hit
    # (src/bwamem_pair.c:376-389): mate B = 126 bases of the novel ALT stretch + 24 bases of a unique primary locus
    rng = np.random.default_rng(13)
    def weak_plus_alt(q):
        p1 = int(rng.integers(200_000, 255_000))
        x = np.concatenate([g[q:q + 126], g[p1:p1 + 24]])
        return x
    for i in range(0, n_reads, 2):
        kind = (i // 2) % 5
        q = 265_100 + int(rng.integers(0, 2300))
        if kind == 0:      # the other mate unalignable
            a, b = rng.integers(0, 4, size=L).astype(np.uint8), weak_plus_alt(q)
        elif kind == 1:    # the other mate a good hit on the primary assembly
            p0 = int(rng.integers(1000, 250_000)); a, b = g[p0:p0 + L].copy(), synth.revcomp(weak_plus_alt(q))
        elif kind == 2:    # both mates inside the novel stretch, a proper pair there
            ins = int(rng.integers(250, 450)); a, b = g[q:q + L].copy(), synth.revcomp(g[q + ins - L:q + ins])
        elif kind == 3:    # both with a weak hit on the primary assembly in front of their ALT hit
            ins = int(rng.integers(250, 450)); a, b = weak_plus_alt(q), synth.revcomp(weak_plus_alt(q + ins - L))
        else:
            continue       # (the pairs of the pe_alt set stay)
        if rng.random() < 0.5: a, b = b, a
        reads[i], reads[i + 1] = a, b
if REPEATS and not PE:                                   # chimeric reads (two loci, either strand): supplementary records and SA tags
    rng = np.random.default_rng(6)
    for i in range(3, n_reads, 25):
        k = int(rng.integers(50, 100)); p1, p2 = (int(x) for x in rng.integers(0, n_genome - L, size=2))
        a, b = g[p1:p1 + k].copy(), g[p2:p2 + L - k].copy()
        if rng.random() < 0.5: a = synth.revcomp(a)
        if rng.random() < 0.5: b = synth.revcomp(b)
        reads[i] = np.concatenate([a, b])
if CONTIGS and PE:                            # some pairs with one mate on either side of a cut
    rng = np.random.default_rng(4)
    cuts = np.cumsum([c[1] for c in CONTIGS])[:-1]
    for i in range(4, n_reads, 40):
        c = int(cuts[(i // 40) % len(cuts)])
        a = g[c - 200:c - 200 + L].copy(); b = synth.revcomp(g[c + 60:c + 60 + L])
        reads[i] = a; reads[i + 1] = b
if CONTIGS and not PE:                        # every eighth read straddles a cut between two sequences
    rng = np.random.default_rng(5)
    cuts = np.cumsum([c[1] for c in CONTIGS])[:-1]
    for i in range(0, n_reads, 8):
        c = int(cuts[i // 8 % len(cuts)]); p = c - int(rng.integers(10, L - 10))
        x = g[p:p + L].copy()
        if i % 24 == 0:                        # some of them with two substitutions, so that parts still seed and align
            for q in rng.integers(20, L - 20, size=2): x[q] = (x[q] + 1) & 3
        reads[i] = x if i % 16 else synth.revcomp(x)
fq = os.path.join(work, "r.fa")
if PE:
    asc = synth.codes_to_ascii(reads)
    with open(fq, "wb") as f:
        for i in range(n_reads):
            f.write(b">p%d\n" % (i // 2)); f.write(asc[i].tobytes()); f.write(b"\n")
else:
    synth.write_fasta_reads(fq, reads)
EXTRA = ["-p"] if PE else []
dump = os.path.join(work, "jobs.bin")
if os.path.exists(dump): os.remove(dump)
sam = os.path.join(work, "o.sam")
with open(sam, "w") as f:
    subprocess.check_call([os.path.join(ROOT, "build", "dropin", "bwa-gasal2"), "gase_aln", "-t", "1", "-l", str(L)] + EXTRA + [prefix, fq], stdout=f,
                          stderr=subprocess.DEVNULL, cwd=work, env=dict(os.environ, BMH_GASAL_DUMP=dump))
raw = np.fromfile(dump, dtype=np.uint8); p = 0; digs = []
while p < raw.size:
    ql, tl, h0 = np.frombuffer(raw[p:p + 12].tobytes(), dtype="<u4"); p += 12
    digs.append(hashlib.sha1(bytes([h0 & 255, h0 >> 8]) + raw[p:p + ql].tobytes() + b"|" + raw[p + ql:p + ql + tl].tobytes()).digest()); p += int(ql) + int(tl)
digs.sort()
as_tag = np.full(n_reads, -1, np.int32)
sam_flag = np.zeros(n_reads, np.int32); sam_pos = np.zeros(n_reads, np.int64); sam_nm = np.full(n_reads, -1, np.int32)
sam_cigar = [""] * n_reads; sam_md = [""] * n_reads
for line in open(sam):
    if line[0] == "@": continue
    c = line.rstrip("\n").split("\t")
    if int(c[1]) & 0x900: continue
    r = int(c[0][1:]) if not PE else 2 * int(c[0][1:]) + (1 if int(c[1]) & 0x80 else 0)
    sam_flag[r] = int(c[1]); sam_pos[r] = int(c[3]); sam_cigar[r] = c[5]
    for tag in c[11:]:
        if tag.startswith("AS:i:"): as_tag[r] = int(tag[5:])
        if tag.startswith("NM:i:"): sam_nm[r] = int(tag[5:])
        if tag.startswith("MD:Z:"): sam_md[r] = tag[5:]
# every SAM line of the default run and of a run with -a (all alignments, secondary ones included): the regions that survive
# mem_sort_dedup_patch, their primary / secondary / supplementary status (mem_mark_primary_se) and MAPQ
def sam_lines(path):
    rows = []
    for line in open(path):
        if line[0] == "@": continue
        c = line.rstrip("\n").split("\t")
        tags = {t[:2]: t[5:] for t in c[11:]}
        rows.append((int(c[0][1:]) if not PE else 2 * int(c[0][1:]) + (1 if int(c[1]) & 0x80 else 0), int(c[1]), int(c[3]), int(c[4]), c[5], int(tags.get("NM", -1)), int(tags.get("AS", -1)), int(tags.get("XS", -1)), tags.get("MD", ""), c[2]))
    return rows
sam_a = os.path.join(work, "o_all.sam")
with open(sam_a, "w") as f:
    subprocess.check_call([os.path.join(ROOT, "build", "dropin", "bwa-gasal2"), "gase_aln", "-a", "-t", "1", "-l", str(L)] + EXTRA + [prefix, fq], stdout=f,
                          stderr=subprocess.DEVNULL, cwd=work)
def pack(rows):
    return dict(read=np.array([r[0] for r in rows], np.int32), flag=np.array([r[1] for r in rows], np.int32), pos=np.array([r[2] for r in rows], np.int64),
                mapq=np.array([r[3] for r in rows], np.int32), cigar=np.array([r[4] for r in rows]), nm=np.array([r[5] for r in rows], np.int32),
                as_=np.array([r[6] for r in rows], np.int32), xs=np.array([r[7] for r in rows], np.int32), md=np.array([r[8] for r in rows]),
                rname=np.array([r[9] for r in rows]))
lines_def = pack(sam_lines(sam)); lines_all = pack(sam_lines(sam_a))
sam_text = np.frombuffer("".join(l for l in open(sam) if l[0] != "@").encode(), dtype=np.uint8)       # the records, verbatim
sam_header = np.frombuffer("".join(l for l in open(sam) if l.startswith("@SQ")).encode(), dtype=np.uint8)
seeds = B.seed_file(prefix, fq, 19)
extra = {}
if ALT:                                       # the genome itself (2 bits per base) and the .alt file, as the test's inputs
    extra = dict(genome_packed=np.packbits(np.stack([g >> 1, g & 1], axis=1).reshape(-1).astype(np.uint8)), alt_file=np.frombuffer(open(prefix + ".alt", "rb").read(), dtype=np.uint8))
np.savez_compressed(os.path.join(ROOT, "gpurun_out", OUT), n_genome=n_genome, genome_seed=42, genome_kw=repr(GENOME_KW), contigs=repr(CONTIGS), reads=reads, **extra,
                    job_digests=np.frombuffer(b"".join(digs), dtype=np.uint8).reshape(-1, 20), as_tag=as_tag,
                    **{"def_" + k: v for k, v in lines_def.items()}, **{"all_" + k: v for k, v in lines_all.items()},
                    sam_text=sam_text, sam_header=sam_header,
                    sam_flag=sam_flag, sam_pos=sam_pos, sam_nm=sam_nm, sam_cigar=np.array(sam_cigar), sam_md=np.array(sam_md),
                    **{k: seeds[k] for k in ("rbeg", "qbeg", "score", "n_ref_pos", "prefix")})
print("wrote", OUT, ":", len(digs), "jobs")
