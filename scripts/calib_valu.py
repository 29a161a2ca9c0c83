#!/usr/bin/env python3
"""bmh_calib_valu sweep: fraction of the guide's integer-VALU figure (256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz = 7.86e13 lane-ops/s, one
wave64 instruction per 2 cycles per SIMD) per instruction kind and occupancy, WITH the placement of every wave (HW_ID / XCC_ID): how
many waves each SIMD of each CU of each XCD received.  Run it plain, or under
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -- python3 scripts/calib_valu.py
usage: calib_valu.py [iters]"""
import collections, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(ROOT, "bwa-mem_gpu_amd"))
import numpy as np, torch, bwamem_hip as B
L = B.load_library(); torch.cuda.init(); torch.zeros(1, device="cuda")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
PEAK = 256 * 4 * 32 * 2.4e9
names = ["independent max/add", "dependent chain", "dependent DPP (+s_nop 1)", "independent DPP", "packed i16 add/max", "bfe_i32", "independent v_fma_f32",
         "independent v_pk_fma_f32", "packed DP mix (pk_mad/sub/max/min/add_u16, perm, and)"]


def decode(words):
    """HW_ID of gfx9: wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]; XCC_ID[3:0] of HW_REG_XCC_ID"""
    hw = words & 0xFFFFFFFF; xcc = (words >> 32) & 0xF
    return xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, (hw >> 4) & 3


for mode, nm in enumerate(names):
    row = []
    for w in (1, 2, 4, 8):
        ms = C.c_float(); ops = C.c_double(); n = C.c_uint(0)
        place = np.zeros(256 * 8 * 4 * 2, np.uint64)
        assert L.bmh_calib_valu_placed(mode, w, iters, None, C.byref(ms), C.byref(ops), place.ctypes.data, C.byref(n)) == 0, B.lib._err(L)
        xcc, se, sh, cu, simd = decode(place[: n.value])
        per_simd = collections.Counter(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist(), simd.tolist()))
        per_cu = collections.Counter(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
        hist = collections.Counter(per_simd.values())
        row.append(f"{w}w {ops.value / (ms.value * 1e-3) / PEAK:.3f} [{len(per_cu)} CUs / {len(per_simd)} SIMDs in {len(set(xcc.tolist()))} XCDs; waves per SIMD: "
                   + ", ".join(f"{k}x{v}" for k, v in sorted(hist.items())) + "]")
    print(f"{nm:28s}", "  ".join(row), flush=True)
print("(fraction 0.5 = one wave64 instruction per 4 cycles per SIMD; waves per SIMD 'k x v' = v SIMDs received k waves)")
