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
work = "/tmp/jobs_golden"; os.makedirs(work, exist_ok=True)
n_genome, n_reads, L = 300_000, 600, 150
g = synth.make_genome(n_genome, seed=42)
idx = fmindex.build_fmd_index(g, device="cuda:0")
prefix = os.path.join(work, "g.fa"); fmindex.write_index(prefix, idx); fmindex.write_bns(prefix, g)
reads, _ = synth.make_reads(g, n_reads, L, seed=21, sub_rate=0.02, indel_frac=0.2)
fq = os.path.join(work, "r.fa"); synth.write_fasta_reads(fq, reads)
dump = os.path.join(work, "jobs.bin")
if os.path.exists(dump): os.remove(dump)
sam = os.path.join(work, "o.sam")
with open(sam, "w") as f:
    subprocess.check_call([os.path.join(ROOT, "build", "dropin", "bwa-gasal2"), "gase_aln", "-t", "1", "-l", str(L), prefix, fq], stdout=f,
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
    r = int(c[0][1:])
    sam_flag[r] = int(c[1]); sam_pos[r] = int(c[3]); sam_cigar[r] = c[5]
    for tag in c[11:]:
        if tag.startswith("AS:i:"): as_tag[r] = int(tag[5:])
        if tag.startswith("NM:i:"): sam_nm[r] = int(tag[5:])
        if tag.startswith("MD:Z:"): sam_md[r] = tag[5:]
seeds = B.seed_file(prefix, fq, 19)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "jobs_golden.npz"), n_genome=n_genome, genome_seed=42, reads=reads,
                    job_digests=np.frombuffer(b"".join(digs), dtype=np.uint8).reshape(-1, 20), as_tag=as_tag,
                    sam_flag=sam_flag, sam_pos=sam_pos, sam_nm=sam_nm, sam_cigar=np.array(sam_cigar), sam_md=np.array(sam_md),
                    **{k: seeds[k] for k in ("rbeg", "qbeg", "score", "n_ref_pos", "prefix")})
print("wrote jobs_golden.npz:", len(digs), "jobs")
