import sys, os, subprocess, io
sys.path.insert(0, "bwa-mem_gpu_amd")
work = sys.argv[1]; threads = sys.argv[2]
exe = os.path.abspath("build/dropin/bwa-gasal2")
prefix = os.path.join(work, "g.fa"); fq = os.path.join(work, "reads.fa")
def ref(t):
    r = subprocess.run([exe, "gase_aln", "-t", t, "-l", "150", "-p", prefix, fq], stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=work)
    return [l for l in r.stdout.decode().split("\n") if l and l[0] != "@"]
from bwamem_hip.aligner import Aligner
al = Aligner(prefix)
def ours():
    buf = io.StringIO(); al.align_file(fq, buf, batch_reads=1 << 30, paired=True)
    return [l for l in buf.getvalue().split("\n") if l and l[0] != "@"]
R = [ref("1"), ref("1"), ref(threads), ref(threads)]
O = [ours(), ours(), ours()]
def nd(a, b): return sum(1 for x, y in zip(a, b) if x != y) + abs(len(a) - len(b))
print("ref vs ref0:", [nd(R[0], x) for x in R])
print("ours vs ours0:", [nd(O[0], x) for x in O])
print("ours0 vs refs:", [nd(O[0], x) for x in R])
names = {x.split("\t")[0] for x, y in zip(O[0], R[0]) if x != y}
os.makedirs("gpurun_out", exist_ok=True)
with open("gpurun_out/det_t1_diff.txt", "w") as f:
    for tag, S in (("OURS", O[0]), ("REF1", R[0])):
        for l in S:
            if l.split("\t")[0] in names:
                f.write(tag + " " + l + "\n")
r = subprocess.run([exe, "gase_aln", "-t", "1", "-l", "150", "-p", prefix, fq], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, cwd=work)
open("gpurun_out/det_ref_stderr.txt", "w").write(r.stderr.decode()[-6000:])
