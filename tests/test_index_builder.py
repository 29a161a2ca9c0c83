"""The FMD index builder writes byte-identical files to the reference's `bwa index` CLI
(golden files built by bwa_index/ with both passes of build_index.sh, tests/golden/make_golden.py)."""
import os

import numpy as np

import common
from bwamem_hip import fmindex, synth


def test_builder_matches_reference_index_files(tmp_path):
    g = synth.make_genome(20011, seed=42)
    idx = fmindex.build_fmd_index(g)
    p = str(tmp_path / "g")
    fmindex.write_index(p, idx)
    fmindex.write_bns(p, g)
    for ext in (".bwt", ".sa", ".pac", ".ann", ".amb"):
        want = open(os.path.join(common.GOLDEN, "ref_index_g20011" + ext), "rb").read()
        got = open(p + ext, "rb").read()
        assert got == want, ext


def test_read_index_roundtrip(tmp_path):
    g = synth.make_genome(50_000, seed=3)
    idx = fmindex.build_fmd_index(g)
    p = str(tmp_path / "g")
    fmindex.write_index(p, idx)
    back = fmindex.read_index(p)
    assert back.primary == idx.primary and back.seq_len == idx.seq_len and back.n_sa == idx.n_sa
    assert np.array_equal(back.bwt_words, idx.bwt_words) and np.array_equal(back.sa, idx.sa)
    assert np.array_equal(back.L2, idx.L2)


def test_golden_fasta_is_the_generator_output():
    fa = open(os.path.join(common.GOLDEN, "ref_index_g20011.fa")).read().split("\n", 1)[1].replace("\n", "")
    g = synth.make_genome(20011, seed=42)
    assert fa == synth.codes_to_ascii(g).tobytes().decode()
