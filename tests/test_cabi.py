"""CPU-side checks of the C-ABI boundary: the library loads without a GPU and exports exactly the
symbols include/mi355seg.h declares, with matching arity in the ctypes table.  No compute calls."""
import ctypes
import os
import re

import pytest

import __graft_entry__ as entry
from rnd_semantic_segmentation_amd import _lib


@pytest.fixture(scope="module")
def built():
    entry.build()
    return _lib.lib()


def _header_decls():
    src = open(_lib.HEADER_PATH).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(?:int|size_t|const char\*)\s+(mi_\w+)\s*\(([^;{]*)\)\s*;", src):
        args = m.group(2).strip()
        decls[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    return decls


def test_header_and_ctypes_table_agree(built):
    decls = _header_decls()
    assert set(decls) == set(_lib.SIGNATURES), set(decls) ^ set(_lib.SIGNATURES)
    for name, nargs in decls.items():
        assert len(_lib.SIGNATURES[name][1]) == nargs, name


def test_library_exports_every_declared_symbol(built):
    h = ctypes.CDLL(_lib.LIB_PATH)
    for name in _header_decls():
        assert hasattr(h, name), name
    assert built.mi_version() == 100


def test_argument_validation_needs_no_gpu(built):
    rc = built.mi_conv_gemm(None, None, None, 1, 1, 1, 64, 1, 1, 4, 1, 1, 0, 1, 0, None, None, None, None, None, 0, 0, 0.0, None)
    assert rc == -22 and b"null operand" in built.mi_last_error()
    assert built.mi_conv_wgrad_workspace(8, 97, 97, 256, 256, 3) > 0
    assert built.mi_upsample_ce_workspace(8, 97, 97, 19, 769, 769) >= 8 * 769 * 97 * 19 * 4


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.MiError, match="no CPU fallback"):
        _lib.lib()
