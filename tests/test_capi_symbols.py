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
    assert lib.rcmarl_abi_version() == 4       # host-only call
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True).stdout
    exported = set(re.findall(r"\bT (rcmarl_\w+)", out))
    assert set(header_functions()) <= exported


def test_mb_job_structure_is_the_same_in_the_header_the_library_and_the_binding():
    """rcmarl_mb_job is declared three times (include/rcmarl.h, csrc/rcmarl_common.h, capi.MbJob): the library reports its own layout,
    the ctypes structure must match it field by field, and the public header must declare the same fields in the same order."""
    import ctypes
    from rcmarl_amd import build, capi
    lib = capi.CLib(build.build_hip())
    fields = [f[0] for f in capi.MbJob._fields_]
    assert lib.rcmarl_mb_job_layout(0) == ctypes.sizeof(capi.MbJob)
    for k, name in enumerate(("x_seed_stride", "theta", "agents", "n_adv", "in_dim", "ldp", "y", "perm", "loss_out", "ovf_flags"), 1):
        assert lib.rcmarl_mb_job_layout(k) == getattr(capi.MbJob, name).offset, name
    assert lib.rcmarl_mb_job_layout(11) == -1
    txt = open(os.path.join(ROOT, "include", "rcmarl.h")).read()
    body = re.search(r"typedef struct rcmarl_mb_job \{(.*?)\} rcmarl_mb_job;", txt, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    decl = re.findall(r"(\w+)\s*[;,]", body)
    assert decl == fields, (decl, fields)


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
        ("rcmarl_minibatch_fit_multi", (None, 3, 1, 5, 100, 20, 128, 32, 10, 0.01, None)),
        ("rcmarl_consensus_params", (None, None, None, None, 1, 5, 64, 40, 4, 1, None, None, None)),
        ("rcmarl_layer1_forward", (None, 0, None, None, 1, 5, 100, 10, 20, 704, 128, None)),
        ("rcmarl_lattice_encode", (None, 0, None, 1, 100, 10, None, 0, 0, None, 0, 0, None, None)),
        ("rcmarl_w1_split", (None, None, None, 1, 5, 10, 20, 704, 1, 1, None)),
        ("rcmarl_layer1_forward_lattice", (None, 0, 0, None, 0, 0, None, None, 1, 5, 100, 10, 20, 704, 128, None)),
        ("rcmarl_layer1_backward_sgd_lattice", (None, 0, 0, None, 0, 0, None, None, None, 1, 5, 100, 10, 20, 704, 0.01, None, 0, 0, None)),
        ("rcmarl_mid_fit_lattice", (None, None, None, None, None, 0, 0, 1, 5, 100, 10, 20, 704, 128, None, None)),
        ("rcmarl_shuffle_perms", (None, None, 1, 1, 100, None, 1, None)),
        ("rcmarl_mid_fit", (None, None, None, None, 1, 5, 100, 10, 20, 704, 128, None)),
        ("rcmarl_mid_value", (None, None, None, 0.9, None, 1, 5, 100, 10, 20, 704, 128, None)),
        ("rcmarl_mid_value_f32", (None, None, None, 0.9, None, 1, 5, 100, 10, 20, 704, 128, None)),
        ("rcmarl_consensus_params_circulant", (None, None, None, 1, 5, 64, 40, 4, 1, None, None, None)),
        ("rcmarl_lattice_pack_dz", (None, None, 1, 5, 100, 20, 128, 1, 4, None)),
        ("rcmarl_lattice_pack_dz_rowsum", (None, None, None, 61, 41, 1, 5, 100, 20, 128, 1, 4, None)),
        ("rcmarl_dense_forward", (None, 0, 0, 1, 10, None, 0, 200, None, 1, 5, 100, 10, 32, 1472, 128, None)),
        ("rcmarl_dense_backward_data", (None, None, 0, None, None, 1, 5, 100, 32, 32, 1472, 128, None)),
        ("rcmarl_dense_backward_sgd", (None, 0, 0, 1, 10, None, None, 0, None, 1, 5, 100, 10, 32, 1472, 128, 0.01, None)),
        ("rcmarl_wide_head_value", (None, None, None, 0.9, None, 1, 5, 100, 10, 32, 1472, 128, None)),
        ("rcmarl_wide_head_fit", (None, None, None, None, None, None, 1, 5, 100, 10, 32, 1472, 128, None)),
        ("rcmarl_wide_bias_grad", (None, None, 1, 5, 100, 32, 128, None)),
        ("rcmarl_wide_small_sgd", (None, None, None, None, None, 1, 5, 100, 10, 32, 1472, 0.01, None)),
        ("rcmarl_wide_consensus_head", (None, None, None, None, None, None, None, None, None, None, None, None, 1, 5, 100, 10, 32,
                                        1472, 128, 4, 1, None)),
        ("rcmarl_wide_head_apply", (None, None, None, 1, 5, 100, 10, 32, 1472, None)),
        ("rcmarl_copy3d", (None, 0, 64, None, 0, 64, 1, 5, 40, None, None)),
        ("rcmarl_layer1_forward_lattice_pk", (None, 0, 0, None, 0, 0, None, None, 0, None, 0, None, 0, None, 1, 5, 100, 10, 128, 1600, None)),
        ("rcmarl_pk_pack_w2", (None, None, None, None, None, 1, 5, 10, 128, 18048, None)),
        ("rcmarl_pk_forward2", (None, None, 2, None, None, None, 2, None, 8, None, None, 1, 5, 100, 10, 128, 18048, 128, None)),
        ("rcmarl_wide_consensus_head_nrm", (None, None, 1, None, None, None, None, None, None, None, None, None, None, 1, 5, 100, 10, 128, 18048,
                                            128, 4, 1, None)),
        ("rcmarl_pk_head", (None, None, None, 0.9, 2, None, None, None, 1, 5, 100, 10, 128, 18048, 128, None)),
        ("rcmarl_pk_backward_data", (None, 2, None, None, None, 8, None, None, 5, 8, None, None, 1, 5, 100, 128, 128, None)),
        ("rcmarl_pk_backward_w2", (None, 8, None, 8, None, None, None, None, None, 1, 5, 100, 10, 128, 18048, 0.01, None)),
        ("rcmarl_pk_small_sgd", (None, None, None, None, None, None, None, None, 1, 5, 100, 10, 128, 18048, 128, 0.01, None)),
    ]
    for name, args in bad:
        with pytest.raises(capi.RcmarlError, match="RCMARL_ERR_ARG|RCMARL_ERR_UNSUPPORTED"):
            getattr(lib, name)(*args)


def test_circulant_kernel_coverage_query():
    """Host-only query: which (N, d, H) the circulant consensus kernel serves (d = 2H+2 with a generated shared network,
    tile image within the LDS); everything else must go through the general entry point."""
    from rcmarl_amd import build, capi
    lib = capi.CLib(build.build_hip())
    q = lib.rcmarl_consensus_params_circulant_supported
    assert q(256, 18, 8) == 1 and q(5, 4, 1) == 1 and q(1024, 66, 32) == 1 and q(64, 10, 4) == 1
    assert q(256, 18, 1) == 0          # d != 2H+2
    assert q(256, 20, 9) == 0          # no generated network
    assert q(3, 4, 1) == 0             # d > N
    assert q(5000, 18, 8) == 0         # [N][16] tile image exceeds the LDS
