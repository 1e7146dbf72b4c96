"""The C-ABI library: loads without a GPU, exports every symbol include/ftx.h declares, and its
argument checks answer before anything is launched (no compute calls here)."""
import ctypes
import os
import re

from fusiontransformer_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ftx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ftx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(ftx_lib):
    syms = declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(ftx_lib, s), s
    assert sorted(_lib.SIGNATURES) == syms, "fusiontransformer_amd/_lib.py SIGNATURES must mirror include/ftx.h"


def test_version_and_capacity(ftx_lib):
    assert ftx_lib.ftx_version() >= 100
    for n in (0, 1, 31, 32, 33, 1000, 81237):
        cap = ftx_lib.ftx_hashtable_capacity(n)
        assert cap >= 64 and cap >= 2 * n and cap & (cap - 1) == 0


def test_argument_errors_are_reported_not_thrown(ftx_lib):
    assert ftx_lib.ftx_hash(None, -1, None, None) == -1
    assert b"n < 0" in ftx_lib.ftx_last_error()
    assert ftx_lib.ftx_hash(None, 0, None, None) == 0          # empty input is a no-op, not an error
    assert ftx_lib.ftx_spconv_pairs_gemm(None, 0, None, None, 0, None, 8, 6, 8, 27, None, None) == -1   # channels not multiple of 4
    assert b"multiples of 4" in ftx_lib.ftx_last_error()
    assert ftx_lib.ftx_hashtable_build(None, 0, None, None, 100, None) == -1               # capacity not a power of two
    assert ftx_lib.ftx_lift_gather_fwd(None, None, None, 4, 1, 24, 24, 95, 370, 1226, None, None) == -1


def test_header_has_no_torch_types():
    text = open(os.path.join(ROOT, "include", "ftx.h")).read()
    assert 'extern "C"' in text
    code = re.sub(r"/\*.*?\*/", "", text, flags=re.S)   # declarations only
    assert "torch" not in code.lower() and "at::" not in code and "Tensor" not in code
