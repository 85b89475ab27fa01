"""Load the hipemu build of the kernel sources (TEST INFRASTRUCTURE)."""
import ctypes
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(_HERE, "hipemu"))
_emu = None


def emu_lib():
    global _emu
    if _emu is None:
        import build_emu
        from rcmarl_amd.capi import CLib
        _emu = CLib(build_emu.build_emu(), needs_hip=False)
    return _emu


def p(arr):
    """Host pointer of a C-contiguous numpy array (or None)."""
    if arr is None:
        return None
    assert arr.flags["C_CONTIGUOUS"]
    return arr.ctypes.data_as(ctypes.c_void_p)


def pad64(n):
    return (n + 63) // 64 * 64
