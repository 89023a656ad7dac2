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


def test_documented_divisibility_rules_are_enforced(built):
    """include/mi355seg.h: mi_conv_gemm needs Ca % 32 == 0 (and, when Ca % 64 != 0, a stride-1 / same-size launch), N % 8 == 0 (N % 16 with sign-bit masks);
    checked before any launch."""
    one = ctypes.c_void_p(16)          # non-null, 16-byte aligned dummy: validation fails before it is dereferenced

    def gemm(N, Ca=64, flags=0, mask_out=None, stride=1, Ha=1):
        return built.mi_conv_gemm(one, one, one, 1, Ha, Ha, Ca, 1, 1, N, 1, stride, 0, 1, 0, None, None, None, None, mask_out, flags, 0, 0.0, None)

    assert gemm(4) == -22 and b"multiple of 8" in built.mi_last_error()
    assert gemm(12) == -22 and b"multiple of 8" in built.mi_last_error()
    assert gemm(8, Ca=48) == -22 and b"multiple of 32" in built.mi_last_error()
    assert gemm(8, Ca=32, stride=2, Ha=2) == -22 and b"not a multiple of 64" in built.mi_last_error()      # 32-channel slabs: the stride-1 main loop only
    rc = built.mi_conv_gemm(one, one, one, 1, 1, 1, 96, 1, 1, 8, 1, 1, 0, 1, 0, None, None, one, None, None, 2, 0, 0.0, None)      # ... and no residual tile
    assert rc == -22 and b"not a multiple of 64" in built.mi_last_error()
    assert gemm(8, flags=64, mask_out=one) == -22 and b"N % 16" in built.mi_last_error()
    # mi_conv_wgrad: out_map 1 is bounded by ncls and by the size of dw
    ws = ctypes.c_void_p(4096)
    rc = built.mi_conv_wgrad(one, one, one, 1, 4, 4, 64, 4, 4, 704, 1, 1, 0, 1, None, 0, 1, 2, 10, ws, 1 << 40, None)
    assert rc == -22 and b"dw holds" in built.mi_last_error()
    rc = built.mi_conv_wgrad(one, one, one, 1, 4, 4, 64, 4, 4, 704, 1, 1, 0, 1, None, 0, 1, 20, 1 << 30, ws, 1 << 40, None)
    assert rc == -22 and b"36*ncls" in built.mi_last_error()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.MiError, match="no CPU fallback"):
        _lib.lib()
