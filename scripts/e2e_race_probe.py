"""Debug aid: run the reference binary (drop-in build) several times at -t N on an existing e2e work directory and
report run-to-run differences plus the shim's result check (BMH_GASAL_CHECK)."""
import sys, os, subprocess
work = sys.argv[1]; threads = sys.argv[2]; n = int(sys.argv[3])
exe = os.path.abspath(os.path.join("build", "dropin", os.environ.get("E2E_EXE", "bwa-gasal2")))      # E2E_EXE=bwa-gasal2-seqidx: scripts/build_dropin.sh
prefix = os.path.join(work, "g.fa"); fq = os.path.join(work, "reads.fa")
env = dict(os.environ, BMH_GASAL_CHECK="1")
env.update({k: v for k, v in (a.split("=", 1) for a in sys.argv[4:] if "=" in a)})
pe = [] if "se" in sys.argv[4:] else ["-p"]
base = None
import hashlib
for k in range(n):
    r = subprocess.run([exe, "gase_aln", "-t", threads, "-l", "150"] + pe + [prefix, fq], stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=work, env=env)
    recs = [l for l in r.stdout.decode().split("\n") if l and l[0] != "@"]
    err = r.stderr.decode()
    chk = [l for l in err.split("\n") if "gasal check" in l or "bwa_gen_cigar2" in l]
    big = sum(1 for l in recs if any(f.startswith("AS:i:") and int(f[5:]) > 1000 for f in l.split("\t")[11:]))
    if base is None:
        base = recs
    d = sum(1 for a, b in zip(base, recs) if a != b)
    print(f"run {k}: rc={r.returncode} records {len(recs)} differ-from-run0 {d} AS>1000 {big} check-lines {len(chk)}", chk[:3], flush=True)
