"""ALT contigs, the parts that need no GPU: the .alt file is read the reference's way (src/bntseq.c:179-200), and the host region tail's
two-round primary marking (mem_mark_primary_se, src/bwamem.c:714-760) on a hand-made read -- the device-resident comparison against the
reference's SAM text is tests/test_gpu_parity.py::test_aligner_writes_reference_sam[alt_golden.npz / pe_alt_golden.npz]."""
import ctypes as C

import numpy as np


def test_alt_file_is_read_like_the_reference(tmp_path):
    from bwamem_hip.aligner import read_alt
    contigs = [("chr1", 1000), ("chr1_alt", 100), ("chrUn", 50), ("HLA-A*01:01", 70)]
    p = str(tmp_path / "g.fa")
    with open(p + ".alt", "wb") as f:
        f.write(b"@SQ\tSN:chr1_alt\tLN:100\n")                       # header lines are skipped (first character '@')
        f.write(b"chr1_alt\t0\tchr1\t101\t60\t100M\t*\t0\t0\t*\t*\n")     # first field = the name
        f.write(b"HLA-A*01:01\r\n")                                 # a bare name, CR LF line end
        f.write(b"unknown_contig\t0\tchr1\t1\t0\t*\n")               # not in the index: ignored
        f.write(b"chrUn")                                           # last line without a terminator: never completed, never looked up (fgetc loop)
    assert read_alt(p, contigs).tolist() == [0, 1, 0, 1]
    assert read_alt(str(tmp_path / "nothing"), contigs).tolist() == [0, 0, 0, 0]


def test_host_tail_marks_primaries_in_two_rounds():
    """Three overlapping hits of one read: the best one on an ALT contig, two on the primary assembly.  Round one (all hits): the ALT hit
    is the parent of both others, so the better primary-assembly hit gets alt_sc = its score; round two (primary assembly only): that
    hit becomes a primary line with the third as its secondary.  Records: [11] = secondary_all, [12] = secondary, [15] = reported |
    is_alt << 1 | alt_sc << 2."""
    import bwamem_hip as B
    from bwamem_hip.lib import ChainOpt, PostOpt
    L = B.load_library()
    l_pac = 4000
    pac = np.zeros(l_pac // 4 + 2, np.uint8)
    ctg_off = np.array([0, 3000], np.int64)                       # sequence 0: primary assembly, sequence 1: ALT
    alt = np.array([0, 1], np.uint8)
    reads = np.zeros(100, np.uint8); read_offs = np.zeros(1, np.uint64)

    def reg(score, qb, qe, rb):
        re = rb + (qe - qb)
        return [0, score, qb, qe, rb & 0xFFFFFFFF, rb >> 32, re & 0xFFFFFFFF, re >> 32]
    regs = np.array([reg(90, 0, 100, 3100), reg(80, 0, 100, 500), reg(70, 0, 100, 1500)], np.int32)
    rpr = np.array([3], np.uint32); fr = np.zeros(1, np.float32)
    co = ChainOpt(); L.bmh_chain_opt_default(C.byref(co))
    po = PostOpt(); L.bmh_post_opt_default(C.byref(po)); po.contig_is_alt = alt.ctypes.data
    assert po.max_XA_hits_alt == 200
    ep = B.ExtParams.default()
    out = np.zeros((3, 16), np.int32); opr = np.zeros(1, np.uint32)
    from bwamem_hip.lib import _i32p, _np_ptr, _u32p, _u64p, _u8p

    def fin():
        return L.bmh_finalize_regs(C.byref(co), C.byref(ep), C.byref(po), l_pac, _np_ptr(pac, _u8p), 1, _np_ptr(reads, _u8p), _np_ptr(read_offs, _u64p), _np_ptr(regs, _i32p),
                                   _np_ptr(rpr, _u32p), fr.ctypes.data_as(C.POINTER(C.c_float)), 2, ctg_off.ctypes.data_as(C.c_void_p), _np_ptr(out, _i32p), _np_ptr(opr, _u32p), 1)
    m = fin()
    assert m == 3 and opr[0] == 3, B.lib._err(L)
    # after the second sort the primary assembly's hits come first: 80, 70, then the ALT hit 90
    assert out[:, 1].tolist() == [80, 70, 90]
    assert out[:, 12].tolist() == [-1, 0, -1]                      # secondary: among the primary assembly's hits; the ALT hit had no parent
    assert out[:, 11].tolist() == [2, 2, -1]                       # secondary_all: both were shadowed by the ALT hit (now at index 2) in round one
    assert (out[:, 15] & 1).tolist() == [1, 0, 1]                  # reported: the primary line and the ALT hit (a supplementary line); not the secondary
    assert ((out[:, 15] >> 1) & 1).tolist() == [0, 0, 1]           # is_alt
    assert (out[:, 15] >> 2).tolist() == [90, 90, 0]               # alt_sc of the hits the ALT one shadowed in round one: the pa tag of the primary line is 80 / 90
    assert out[2, 14] & 0x800 and not (out[0, 14] & 0x900)
    # without the table: one round, the best hit (on what is now an ordinary sequence) is the only primary line
    po.contig_is_alt = None
    m = fin()
    assert m == 3 and out[:, 1].tolist() == [90, 80, 70] and out[:, 12].tolist() == [-1, 0, 0] and out[:, 15].tolist() == [1, 0, 0]
