"""CPU-side checks of the C-ABI library: it loads, and exports every symbol include/indextts_hip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "indextts_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(itts_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from indextts import _native
    if not os.path.exists(_native.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_native.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 13
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/indextts_hip.h but not exported"
    assert set(syms) == set(_native.EXPORTED_SYMBOLS)
    # the diagnostic entry points (include/indextts_hip_diag.h: tuning overrides, time stamps) are NOT in the product library
    diag = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "indextts_hip_diag.h")).read(), flags=re.S)
    dsyms = sorted(set(re.findall(r"\b(itts_[a-z0-9_]+)\s*\(", diag)))
    assert dsyms == ["itts_debug_set", "itts_debug_stamps", "itts_debug_stamps_conv", "itts_debug_stamps_sample"]
    for s in dsyms:
        assert not hasattr(lib, s), f"{s} (diagnostic build only) is exported by the product library"
    lib.itts_abi_version.restype = ctypes.c_int
    assert lib.itts_abi_version() == 8
    lib.itts_packed_bytes.restype = ctypes.c_int64
    assert lib.itts_packed_bytes(1, 1280, 3840, 1) == 1280 * 3840 * 2
    assert lib.itts_packed_bytes(7, 24, 1, 0) == 7 * 1 * 2 * 1024


def test_invalid_arguments_are_reported_not_launched():
    from indextts import _native
    L = _native.lib()
    a = _native.ConvArgs()
    rc = L.itts_gemm_conv(ctypes.byref(a), None)
    assert rc == 1 and b"null pointer" in L.itts_last_error()
    s = _native.SkinnyArgs()
    assert L.itts_gemm_skinny(ctypes.byref(s), None) == 1


def test_no_cpu_fallback():
    import torch
    from indextts import _native
    with pytest.raises(_native.NativeError):
        _native.aa_snake(torch.zeros(1, 4, 8), torch.zeros(8), torch.zeros(8), [0.0] * 12, [0.0] * 12)
