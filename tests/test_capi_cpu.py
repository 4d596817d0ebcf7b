"""CPU: the C-ABI library loads, exports every symbol include/mipgen_accel.h declares, and refuses to compute
without a GPU (there is no CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re

import pytest

from mipgen_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mipgen_accel.h")).read()
    return sorted(set(re.findall(r"\b(mipgen_accel_[a-z_]+)\s*\(", text)))


def test_header_symbols_all_exported():
    assert os.path.exists(capi.LIB_PATH), "build with: python -c 'import __graft_entry__ as g; g.build()'"
    lib = C.CDLL(capi.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 20
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert sorted(capi.EXPORTED_SYMBOLS) == declared


def test_abi_version_and_struct_sizes():
    lib = capi.load_library()
    assert lib.mipgen_accel_abi_version() == capi.ABI_VERSION == 6
    # layout agreed between ctypes and the C header (spot checks that catch padding mistakes)
    assert C.sizeof(capi.Grid) == 32
    assert C.sizeof(capi.Candidate) == 24
    assert C.sizeof(capi.Survivor) == 24
    assert C.sizeof(capi.CandidateInts) == 80
    assert C.sizeof(capi.Region) == 6 * 4 + 5 * 8 + 44 * 8
    assert C.sizeof(capi.Params) == 7 * 4 + 2 * 256 * 4 + 2 * 4 + 4 + 3 * 8 + 8 * 4


def test_no_cpu_fallback_without_device():
    lib = capi.load_library()
    if lib.mipgen_accel_device_count() > 0:
        pytest.skip("a GPU is present")
    p = capi.make_params(152, 162)
    h = C.c_void_p()
    rc = lib.mipgen_accel_create(C.byref(p), 0, None, C.byref(h))
    assert rc == -2                                      # MIPGEN_E_NODEVICE
    assert b"no CPU path" in lib.mipgen_accel_last_error()
    with pytest.raises(capi.AccelError):
        capi.Accel(p)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under mipgen_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "mipgen_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".c", ".hpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                for needle in ("pyoracle", "mipgen_oracle", "import oracle", "from oracle", "libmipgen_refdrv", "mipgen_ref"):
                    assert needle not in text, (os.path.join(dirpath, f), needle)
