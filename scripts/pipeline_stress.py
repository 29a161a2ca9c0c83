"""Threading stress of bmh_aligner_run / bmh_aligner_run_fasta: many tiny batches, several lane counts, streamed and loaded files -- every run must give the same bytes."""
import os, sys, io, hashlib
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "bwa-mem_gpu_amd"))
import numpy as np
import bwamem_hip as B
from bwamem_hip import fmindex, synth
from bwamem_hip.aligner import Aligner
import tempfile
d = tempfile.mkdtemp()
g = synth.make_genome(500_000, seed=8, repeat_frac=0.25)
prefix = os.path.join(d, "g.fa")
fmindex.write_index(prefix, fmindex.build_fmd_index(g)); fmindex.write_bns(prefix, g, contigs=[("a", 200_000), ("b", 300_000)])
for pe in (False, True):
    n = 6000
    reads = (synth.make_pairs(g, n // 2, 120, seed=2) if pe else synth.make_reads(g, n, 120, seed=2))[0]
    fq = os.path.join(d, "r%d.fa" % pe)
    with open(fq, "wb") as f:
        for i, r in enumerate(synth.codes_to_ascii(reads)):
            f.write(b">%s%d\n" % (b"p" if pe else b"r", i // 2 if pe else i) + r.tobytes() + b"\n")
    al = Aligner(prefix, n_threads=4)
    hs = set()
    for it in range(40):
        for lanes, br, stream in ((4, 50, "1"), (3, 50, "0"), (2, 2000, "1"), (5, 10, "1")):
            os.environ["BMH_ALIGNER_LANES"] = str(lanes); os.environ["BMH_ALIGNER_STREAM"] = stream
            buf = io.BytesIO()
            al.align_file(fq, buf, batch_reads=br, paired=pe)
            if pe and br != 50: continue        # (pairs: other batch sizes, other insert-size statistics)
            hs.add(hashlib.sha256(buf.getvalue()).hexdigest())
    print("paired" if pe else "single", "distinct outputs over 40 x 4 runs:", len(hs), flush=True)
    assert len(hs) == 1
    al.close()
print("stress ok")
