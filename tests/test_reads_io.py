"""bmh_reads_load_fasta (csrc/reads_io.cpp): the library's read-file loader against the numpy parse of the same layout."""
import numpy as np
import pytest

from bwamem_hip.aligner import _NT4, read_fasta_reads, read_fasta_reads_numpy


def _write(path, rng, n, crlf=False, blanks=False, tail_nl=True, words=False, readno=False):
    alphabet = np.frombuffer(b"ACGTNacgtnRY", dtype=np.uint8)
    with open(path, "wb") as f:
        for i in range(n):
            s = rng.choice(alphabet, size=int(rng.integers(1, 300))).tobytes()
            nm = b"r%d" % i + ((b"/%d" % (1 + i % 2) if i % 4 else b"/x") if readno else b"") + ((b"\tdesc x" if i % 3 == 0 else b" more words") if words else b"")
            e = b"\r\n" if crlf else b"\n"
            f.write(b">" + nm + e)
            if blanks and i % 5 == 0:
                f.write(e)
            f.write(s + (e if (tail_nl or i < n - 1) else b""))


@pytest.mark.parametrize("kw", [dict(), dict(crlf=True), dict(blanks=True, words=True), dict(tail_nl=False),
                                dict(crlf=True, blanks=True, words=True, tail_nl=False), dict(readno=True), dict(readno=True, words=True, crlf=True)])
def test_loader_equals_numpy_parse(tmp_path, kw):
    """letters, nt4 codes, offsets, lengths and names of files with LF / CR LF line ends, blank lines, descriptions behind the
    name, no newline at the end; 1 read, a few, and enough for the loader's chunks (a file beyond 1 MB is cut at headers and
    parsed by several host threads)"""
    rng = np.random.default_rng(3)
    p = str(tmp_path / "r.fa")
    for n in (1, 7, 20000):
        _write(p, rng, n, **kw)
        a, b = read_fasta_reads(p), read_fasta_reads_numpy(p)
        assert len(a) == len(b) == n
        for k in ("ascii", "offs", "lens", "name_blob", "name_off"):
            assert np.array_equal(getattr(a, k), getattr(b, k)), (kw, n, k)
        if kw.get("readno"):           # "r5/2" is read r5 (trim_readno, src/bwa.c:27-31); "r4/x" keeps its tail
            names = bytes(a.name_blob).split(b"\0")[:n]
            assert names == [b"r%d" % i + (b"" if i % 4 else b"/x") for i in range(n)], names[:6]
        assert np.array_equal(a.codes, _NT4[a.ascii])
        s = a.slice(n // 2, n)
        assert np.array_equal(s.codes, _NT4[s.ascii]) and len(s) == n - n // 2


def test_loader_refuses_what_the_reference_layout_excludes(tmp_path):
    p = str(tmp_path / "bad.fa")
    for text in (b">a\n>b\nACGT\n", b"ACGT\n>a\nACGT\n", b">a\nAC\nGT\n", b">a\nACGT\n>b\n"):
        open(p, "wb").write(text)
        for fn in (read_fasta_reads, read_fasta_reads_numpy):
            with pytest.raises(ValueError, match="alternating"):
                fn(p)
    open(p, "wb").write(b"")
    assert len(read_fasta_reads(p)) == 0
    with pytest.raises(RuntimeError, match="cannot open"):
        read_fasta_reads(str(tmp_path / "missing.fa"))
