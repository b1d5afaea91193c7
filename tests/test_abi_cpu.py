"""The C-ABI library loads on a GPU-less host and exports every symbol include/thinkdiff_hip.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "thinkdiff-mlre_amd", "lib", "libthinkdiff_hip.so")
HDR = os.path.join(ROOT, "include", "thinkdiff_hip.h")


def _declared():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(td_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(LIB), "build first: make -C thinkdiff-mlre_amd (or __graft_entry__.build())"
    lib = ctypes.CDLL(LIB)
    names = _declared()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.td_abi_version() >= 1


def test_python_binding_declares_only_real_symbols():
    import importlib
    hip = importlib.import_module("thinkdiff._hip")
    lib = hip.lib()
    assert lib.td_last_error() is not None


def test_argument_errors_are_reported_without_a_gpu():
    lib = ctypes.CDLL(LIB)
    lib.td_last_error.restype = ctypes.c_char_p
    # K not a multiple of 64 -> TD_ERR_INVALID before any HIP call
    rc = lib.td_linear_bf16(None, 40, None, None, None, 8, 4, 8, 40, 0, None, None, 0, None)
    assert rc == 2 and b"K=40" in lib.td_last_error()
