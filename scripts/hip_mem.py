#!/usr/bin/env python3
"""hip_mem.py [-p] <index prefix> <reads.fa> > out.sam   -- alignment on the device-resident path (bwamem_hip.aligner);
-p: the file holds interleaved pairs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bwa-mem_gpu_amd"))
from bwamem_hip.aligner import Aligner
paired = "-p" in sys.argv
args = [x for x in sys.argv[1:] if x != "-p"]
if len(args) < 2:
    sys.exit(__doc__)
a = Aligner(args[0])
n = a.align_file(args[1], sys.stdout, batch_reads=int(args[2]) if len(args) > 2 else 500_000, paired=paired)
sys.stderr.write(f"[hip_mem] {n} reads\n")
