#!/usr/bin/env python3
"""bmh_calib_valu sweep: fraction of the integer-VALU peak (256 CUs x 4 SIMD-32 x 2.4 GHz = 7.86e13 lane-ops/s) per instruction kind and occupancy"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(ROOT, "bwa-mem_gpu_amd"))
import torch, bwamem_hip as B
L = B.load_library(); torch.cuda.init(); torch.zeros(1, device="cuda")
PEAK = 256 * 4 * 32 * 2.4e9
names = ["independent max/add", "dependent chain", "dependent DPP (+s_nop 1)", "independent DPP", "packed i16 add/max", "bfe_i32", "independent v_fma_f32"]
for mode, nm in enumerate(names):
    row = []
    for w in (1, 2, 4, 8):
        ms = C.c_float(); ops = C.c_double()
        assert L.bmh_calib_valu(mode, w, 5000, None, C.byref(ms), C.byref(ops)) == 0
        row.append(f"{w}w {ops.value / (ms.value * 1e-3) / PEAK:.3f}")
    print(f"{nm:28s}", "  ".join(row))
