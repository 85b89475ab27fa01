"""The C-ABI library builds for gfx950 without a GPU, loads, and exports every
symbol include/rcmarl.h declares; the ctypes table matches the header.  No
compute calls (CPU-only)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    txt = open(os.path.join(ROOT, "include", "rcmarl.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\bint\s+(rcmarl_\w+)\s*\(", txt)))


def test_header_matches_binding_table():
    from rcmarl_amd import capi
    assert header_functions() == sorted(capi.SIGNATURES)


def test_hip_library_builds_loads_and_exports_all_symbols():
    from rcmarl_amd import build, capi
    path = build.build_hip()
    lib = capi.CLib(path)                      # binds every symbol, raises if one is missing
    assert lib.rcmarl_abi_version() == 1       # host-only call
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True).stdout
    exported = set(re.findall(r"\bT (rcmarl_\w+)", out))
    assert set(header_functions()) <= exported


def test_missing_library_fails_loudly(tmp_path):
    from rcmarl_amd import capi
    with pytest.raises(capi.RcmarlError, match="no CPU fallback|not found"):
        capi.CLib(str(tmp_path / "nope.so"))
