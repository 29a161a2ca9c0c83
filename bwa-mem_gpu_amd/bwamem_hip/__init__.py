"""bwamem_hip -- host-side Python mirror of the MI355X seed-and-extend library.

The product is libbwamem_hip.so (hand-written HIP for gfx950, C ABI declared in
include/bwamem_hip.h and include/seed_gen.h).  This package only binds it with
ctypes for tests, bench.py and the multi-GPU launcher, and carries the
synthetic-data and index-building tooling.  There is no CPU implementation of
the hot path in here: if the shared library or a HIP device is missing, calls
raise.
"""
from . import fmindex, synth  # noqa: F401
from .lib import (ExtParams, HipLibraryMissing, Index, SeedWorkspace, extend_batch, lib_path,  # noqa: F401
                  load_library, seed_file)
