"""CPU: the C-ABI library loads and exports every symbol include/modex_hip.h declares, and the
ctypes binding table matches the header (no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "modex_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.findall(r"\bint\s+(mx_\w+)\s*\(", text)


def header_arg_counts():
    text = open(os.path.join(ROOT, "include", "modex_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for name, args in re.findall(r"\bint\s+(mx_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        args = args.strip()
        out[name] = 0 if args in ("", "void") else len(args.split(","))
    return out


@pytest.fixture(scope="module")
def so_path():
    from mod_extraction_amd import build
    return build.build(verbose=False)


def test_library_exports_every_declared_symbol(so_path):
    lib = ctypes.CDLL(so_path)
    syms = header_symbols()
    assert len(syms) >= 4
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/modex_hip.h but not exported"
    assert lib.mx_abi_version() >= 1


def test_binding_table_matches_header(so_path):
    from mod_extraction_amd import _hip
    counts = header_arg_counts()
    assert set(counts) == set(_hip.SIGNATURES), set(counts) ^ set(_hip.SIGNATURES)
    for name, n in counts.items():
        assert len(_hip.SIGNATURES[name]) == n, name
    assert _hip.load().mx_abi_version() == _hip.ABI_VERSION


def test_product_has_no_cpu_fallback():
    import torch
    from mod_extraction_amd import _hip, fx
    with pytest.raises(_hip.HipLibraryError):
        fx.MonoFlangerChorusModule(1, 1, 64, 44100, 1.0, 10.0)(torch.zeros(1, 1, 64), torch.zeros(1, 64))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "mod_extraction_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(d, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "liboracle_ref" not in src, f


def test_entry_points_reject_bad_arguments_before_touching_the_device(so_path):
    """Every entry point validates its arguments first and returns MX_ERR_ARG / MX_ERR_UNSUPPORTED (never throws, never
    launches): NULL pointers and non-positive sizes are refused here on a box without a GPU."""
    from mod_extraction_amd import _hip
    lib = _hip.load()
    zeros = {ctypes.c_void_p: None, ctypes.c_int64: 0, ctypes.c_int32: 0, ctypes.c_float: 0.0, ctypes.c_double: 0.0,
             ctypes.c_uint64: 0, ctypes.c_uint32: 0}
    skip = {"mx_abi_version"}
    checked = 0
    for name, argtypes in _hip.SIGNATURES.items():
        if name in skip:
            continue
        rc = getattr(lib, name)(*[zeros[t] for t in argtypes])
        assert rc in (-1, -2), (name, rc)                     # MX_ERR_ARG or MX_ERR_UNSUPPORTED
        checked += 1
    assert checked >= 40
