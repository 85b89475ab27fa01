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


def test_argument_validation_needs_no_gpu():
    """Every entry point validates its arguments before touching the HIP runtime: bad calls come back as
    RCMARL_ERR_ARG / RCMARL_ERR_UNSUPPORTED (raised as RcmarlError by the binding) even on a machine without a GPU."""
    from rcmarl_amd import build, capi
    lib = capi.CLib(build.build_hip())
    bad = [
        ("rcmarl_consensus_params", (None, None, None, None, 1, 5, 64, 40, 4, 1, None, None, None)),
        ("rcmarl_layer1_forward", (None, 0, None, None, 1, 5, 100, 10, 20, 704, 128, None)),
        ("rcmarl_lattice_encode", (None, 0, None, 1, 100, 10, None, 0, 0, None, 0, 0, None, None)),
        ("rcmarl_w1_split", (None, None, None, 1, 5, 10, 20, 704, 1, 1, None)),
        ("rcmarl_layer1_forward_lattice", (None, 0, 0, None, 0, 0, None, None, 1, 5, 100, 10, 20, 704, 128, None)),
        ("rcmarl_layer1_backward_sgd_lattice", (None, 0, 0, None, 0, 0, None, None, None, 1, 5, 100, 10, 20, 704, 0.01, None, 0, 0, None)),
        ("rcmarl_mid_fit_lattice", (None, None, None, None, None, 0, 0, 1, 5, 100, 10, 20, 704, 128, None)),
        ("rcmarl_shuffle_perms", (None, None, 1, 1, 100, None, 1, None)),
        ("rcmarl_mid_fit", (None, None, None, None, 1, 5, 100, 10, 20, 704, 128, None)),
    ]
    for name, args in bad:
        with pytest.raises(capi.RcmarlError, match="RCMARL_ERR_ARG|RCMARL_ERR_UNSUPPORTED"):
            getattr(lib, name)(*args)
