#!/usr/bin/env python3
"""hip_mem.py <index prefix> <reads.fa> > out.sam   -- single-end alignment on the device-resident path (bwamem_hip.aligner)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bwa-mem_gpu_amd"))
from bwamem_hip.aligner import Aligner
if len(sys.argv) < 3:
    sys.exit(__doc__)
a = Aligner(sys.argv[1])
n = a.align_file(sys.argv[2], sys.stdout, batch_reads=int(sys.argv[3]) if len(sys.argv) > 3 else 500_000)
sys.stderr.write(f"[hip_mem] {n} reads\n")
