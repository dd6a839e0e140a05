"""ctypes loader for oracle/_build/liboracle_ref.so (TEST INFRASTRUCTURE ONLY).

The library is built from oracle/csrc/oracle_ref.c by ``make -C oracle`` (also run by
``__graft_entry__.build()``).  It is never loaded by the product package.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_ref.so")
_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.isfile(_SO):
            subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
        _lib = ctypes.CDLL(_SO)
        f32p, i64p, i64, f64 = (ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int64),
                                ctypes.c_int64, ctypes.c_double)
        _lib.orc_flanger.argtypes = [f32p] * 8 + [i64, i64, i64, f32p, i64p, f32p]
        _lib.orc_flanger.restype = None
        _lib.orc_phaser.argtypes = [f32p] * 6 + [i64, i64, f64, f32p, f32p]
        _lib.orc_phaser.restype = None
    return _lib


def fptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)) if a is not None else None


def iptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)) if a is not None else None
