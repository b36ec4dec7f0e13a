"""CPU tests of the C-ABI boundary: the library builds/loads and exports every symbol include/gdhip.h declares."""

import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "gdhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gd_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    from getdist_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        _lib.build_native()
    lib = _lib.load_library()
    syms = declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), "libgdhip.so does not export %s" % s
        assert s in _lib.SIGNATURES, "no ctypes prototype for %s" % s
    assert set(_lib.SIGNATURES) == set(syms)
    assert b"gfx950" in lib.gd_version()


def test_no_gpu_fails_loudly():
    from getdist_amd import _lib

    lib = _lib.load_library()
    if lib.gd_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(RuntimeError):
        _lib.Context(0)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "getdist_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in re.sub(r"#.*", "", src).replace("kde_oracle", "oracle") or \
                    not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f
