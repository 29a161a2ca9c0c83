"""The C-ABI library loads, exports every symbol include/*.h declares, and fails loudly without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import common

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions(header: str):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    src = re.sub(r"typedef\s+[^;{}()]*\(\s*\*\s*\w+\s*\)\s*\([^;{}]*\)\s*;", "", src)      # function-pointer typedefs are not functions
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{}]*\)\s*;", src)
    return sorted(set(n for n in names if n not in ("defined", "__attribute__")))


def test_library_exports_every_declared_symbol():
    import bwamem_hip as B
    L = B.load_library()
    decl = []
    for h in sorted(os.listdir(os.path.join(ROOT, "include"))):
        if h.endswith(".h"):
            decl += _declared_functions(h)
    assert "seed_gpu" in decl and "bmh_seed_batch" in decl and "bmh_extend_batch" in decl
    missing = [n for n in decl if not hasattr(L, n)]
    assert not missing, f"declared in include/*.h but not exported: {missing}"


def test_python_symbol_list_matches_headers():
    from bwamem_hip.lib import EXPORTED_SYMBOLS
    decl = set()
    for h in os.listdir(os.path.join(ROOT, "include")):
        if h.endswith(".h"):
            decl |= set(_declared_functions(h))
    assert decl <= set(EXPORTED_SYMBOLS) | {"bmh_set_error"}, sorted(decl - set(EXPORTED_SYMBOLS))


def test_no_cpu_fallback_without_device():
    """Without a HIP device the product path reports an error instead of computing on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import bwamem_hip as B
    L = B.load_library()
    assert L.bmh_device_count() == 0
    g, idx = common.genome_and_index(20_000, seed=1)
    with pytest.raises(RuntimeError):
        B.Index.upload(idx)
    with pytest.raises(RuntimeError):
        B.SeedWorkspace(16, 4096)


def test_product_sources_never_reference_the_oracle():
    """include/ and bwa-mem_gpu_amd/ must not import, link or call anything under oracle/."""
    bad = []
    for base in ("include", "bwa-mem_gpu_amd"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".h", ".hip", ".cpp", ".c", ".py")) or f == "Makefile":
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"oracle_py|liboracle|fmd_oracle|ksw_oracle|libref", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad
