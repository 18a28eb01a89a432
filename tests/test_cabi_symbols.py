"""CPU-side checks of the drop-in boundary: the shared library loads, exports every symbol that
include/zotk.h declares, the ctypes signature table covers the same set, and the product package
never reaches into oracle/ (no compute calls here: there is no GPU in this container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "zotk.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(zk_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_path():
    syms = declared_symbols()
    for need in ("zk_create", "zk_encode", "zk_sort_keys", "zk_rle", "zk_kmerize", "zk_hist", "zk_union_sum",
                 "zk_merge_n", "zk_project_dedupe", "zk_split", "zk_trim", "zk_subsample"):
        assert need in syms


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from zotmer_amd import native
    lib = ctypes.CDLL(native.LIB_PATH)
    for s in declared_symbols():
        assert hasattr(lib, s), "libzotk.so does not export %s" % s
    assert sorted(native.SIGNATURES) == declared_symbols()
    native.load()   # binds restype/argtypes for all of them


def test_no_gpu_fails_loudly():
    from zotmer_amd import native
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(native.ZotkError):
        native.Context(0)


def test_product_never_touches_the_oracle():
    """Nothing under zotmer_amd/ imports, includes, links or dlopens anything under oracle/."""
    pats = [r"^\s*import\s+oracle", r"^\s*from\s+oracle", r"#\s*include\s*[<\"][^>\"]*oracle", r"libzkoracle", r"zkoracle\."]
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "zotmer_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")) or f == "Makefile":
                s = open(os.path.join(d, f), errors="replace").read()
                if any(re.search(p, s, flags=re.M) for p in pats):
                    bad.append(f)
    assert not bad, bad
