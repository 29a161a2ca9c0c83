"""gase_aln command-line options -> option structs of the C ABI (Aligner.set_options mirrors src/fastmap.c:166-262)."""
import ctypes as C

import pytest

from bwamem_hip.aligner import Aligner
from bwamem_hip.lib import ChainOpt, ExtParams, PeOpt, PostOpt, load_library


class _Opts:
    """the option state of an Aligner without its device side"""
    def __init__(self):
        L = load_library()
        self.copt = ChainOpt(); L.bmh_chain_opt_default(C.byref(self.copt))
        self.ep = ExtParams.default()
        self.po = PostOpt(); L.bmh_post_opt_default(C.byref(self.po))
        self.pe = PeOpt(); L.bmh_pe_opt_default(C.byref(self.pe))


def test_defaults_are_mem_opt_init():
    o = _Opts()                                           # src/bwamem.c:101-146
    assert (o.copt.a, o.copt.b, o.copt.o_del, o.copt.e_del, o.copt.o_ins, o.copt.e_ins) == (1, 4, 6, 1, 6, 1)
    assert (o.copt.w, o.copt.min_seed_len, o.copt.max_occ, o.copt.max_chain_gap, o.copt.min_chain_weight) == (300, 19, 500, 10000, 0)
    assert abs(o.copt.mask_level - 0.5) < 1e-7 and abs(o.copt.drop_ratio - 0.5) < 1e-7
    assert (o.ep.zdrop, o.ep.end_bonus) == (0, 5)
    assert (o.po.T, o.po.flag_all, o.po.max_XA_hits, o.po.mapQ_coef_fac, o.po.no_multi, o.po.softclip) == (30, 0, 5, 3, 0, 0)
    assert (o.pe.pen_unpaired, o.pe.max_ins, o.pe.max_matesw, o.pe.no_rescue, o.pe.no_pairing) == (17, 10000, 50, 0, 0)


def test_set_options_maps_every_modelled_flag():
    o = _Opts()
    Aligner.set_options(o, "-k 23 -w 50 -c 300 -D 0.4 -G 500 -N 10 -W 25 -X 0.3 -A 2 -B 8 -O 12,13 -E 2,3 -T 60 -h 3 -Q 30 -U 34 -m 20 "
                           "-a -M -Y -S -P -t 16 -K 1000000 -l 150 -d 100 -L 7".split())
    c, e, p, q = o.copt, o.ep, o.po, o.pe
    assert (c.min_seed_len, c.w, c.max_occ, c.max_chain_gap, c.max_chain_extend, c.min_chain_weight) == (23, 50, 300, 500, 10, 25)
    assert abs(c.drop_ratio - 0.4) < 1e-6 and abs(c.mask_level - 0.3) < 1e-6
    assert (c.a, c.b, c.o_del, c.o_ins, c.e_del, c.e_ins) == (2, 8, 12, 13, 2, 3)
    assert (e.a, e.b, e.o_del, e.o_ins, e.e_del, e.e_ins, e.zdrop, e.end_bonus) == (2, 8, 12, 13, 2, 3, 0, 5)      # -d / -L do not reach the GPU extension
    assert (p.T, p.max_XA_hits, p.flag_all, p.no_multi, p.softclip) == (60, 3, 1, 1, 1) and p.mapQ_coef_len == 30.0 and p.mapQ_coef_fac == 3
    assert (q.pen_unpaired, q.max_matesw, q.no_rescue, q.no_pairing) == (34, 20, 1, 1)


def test_set_options_rejects_what_is_not_modelled():
    o = _Opts()
    for bad in (["-x", "pacbio"], ["-r", "1.2"], ["-R", "RG\\tID:x"], ["-R", "@RG\\tSM:x"], ["-R", "@RG\\tID:" + "x" * 256], ["-C", "1"], ["-k"]):
        with pytest.raises(ValueError):
            Aligner.set_options(o, bad)


def test_read_group_and_ignore_alt_options():
    """-R: the line unescaped like bwa_escape, the ID (up to the next tab) for the records' RG:Z tag (bwa_set_rg, src/bwa.c:425-452); -j: the ALT flags dropped"""
    import numpy as np
    o = _Opts()
    o.alt = np.array([0, 1, 1], np.uint8); o.has_alt = True
    o.copt.contig_is_alt = o.alt.ctypes.data; o.po.contig_is_alt = o.alt.ctypes.data
    Aligner.set_options(o, ["-R", "@RG\\tID:grp.1\\tSM:s 1\\tPL:x", "-j"])
    assert o.rg_line == "@RG\tID:grp.1\tSM:s 1\tPL:x" and o.po.rg_id == b"grp.1"
    assert not o.has_alt and not o.alt.any() and not o.copt.contig_is_alt and not o.po.contig_is_alt
    o.contigs = [("a", 5), ("b", 7)]
    assert Aligner.header(o) == "@SQ\tSN:a\tLN:5\n@SQ\tSN:b\tLN:7\n@RG\tID:grp.1\tSM:s 1\tPL:x\n"


def test_read_group_line_is_unescaped_in_one_pass_like_bwa_escape():
    """bwa_escape (src/bwa.c:409-423) walks the line once: "\\\\t" is a backslash followed by 't' (not a backslash and a tab), an unknown escape
    drops both characters, and the ID ends at the first tab OR newline (src/bwa.c:440-446).  -j also forgets a cached chain workspace (its ALT table)."""
    o = _Opts()
    Aligner.set_options(o, ["-R", "@RG\\tID:a\\\\tb\\qc\\nSM:x"])
    assert o.rg_line == "@RG\tID:a\\tbc\nSM:x" and o.po.rg_id == b"a\\tbc"

    class _Ws:
        freed = False
        def free(self): self.freed = True
    import numpy as np
    o = _Opts(); o.alt = np.zeros(2, np.uint8); o.has_alt = True
    ws = _Ws(); o._cw_cache = (ws, 1, 1)
    Aligner.set_options(o, ["-j"])
    assert ws.freed and o._cw_cache is None
