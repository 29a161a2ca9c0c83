#!/usr/bin/env python3
"""hip_mem.py [-p] [gase_aln options] <index prefix> <reads.fa> [batch reads] > out.sam   -- alignment on the device-resident
path (bwamem_hip.aligner); -p: the file holds interleaved pairs; options: see Aligner.set_options (-k -w -c -D -G -N -W -X
-A -B -O -E -T -h -Q -U -m -R -a -M -Y -S -P -j)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bwa-mem_gpu_amd"))
from bwamem_hip.aligner import Aligner
paired = "-p" in sys.argv
argv = [x for x in sys.argv[1:] if x != "-p"]
opts, args, i = [], [], 0
while i < len(argv):
    if argv[i].startswith("-") and not args:
        k = 1 if argv[i] in ("-a", "-M", "-Y", "-S", "-P", "-j") else 2
        opts += argv[i:i + k]; i += k
    else:
        args.append(argv[i]); i += 1
if len(args) < 2:
    sys.exit(__doc__)
a = Aligner(args[0])
a.set_options(opts)
n = a.align_file(args[1], sys.stdout, batch_reads=int(args[2]) if len(args) > 2 else 500_000, paired=paired)
sys.stderr.write(f"[hip_mem] {n} reads\n")
