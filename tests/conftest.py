import os
import sys

import pytest

_HERE = os.path.dirname(os.path.abspath(__file__))
_REPO = os.path.dirname(_HERE)
for p in (_REPO, _HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(_HERE, "golden", "reference_golden.npz"), allow_pickle=False)


def pytest_collection_modifyitems(config, items):
    """`gpu` tests need a real MI355X: on a host without one they are SKIPPED (a plain
    `pytest tests` then runs the CPU suite instead of failing in fixtures)."""
    import torch
    if torch.cuda.is_available():
        return                                # on a GPU box nothing is skipped: a missing librcmarl_hip.so must FAIL the tests
    skip = pytest.mark.skip(reason="needs an MI355X (torch.cuda.is_available() is False)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture
def lattice_form(monkeypatch):
    """Switch the operand form of the lattice path inside one process: the library reads RCMARL_LAT_F16 once, so the switch goes
    through rcmarl_lattice_set_f16_mode(); the environment variable is set too (host-side helpers read it).  Restored afterwards."""
    done = []

    def set_form(bk, mode):
        done.append((bk.lib, bk.lib.rcmarl_lattice_f16_mode()))
        monkeypatch.setenv("RCMARL_LAT_F16", str(mode))
        bk.lib.rcmarl_lattice_set_f16_mode(int(mode))
    yield set_form
    for lib, prev in reversed(done):
        lib.rcmarl_lattice_set_f16_mode(prev)


@pytest.fixture
def wide_form():
    """Switch the dense layers of the wide path between the 16-bit matrix core (1, the default) and the fp32-input MFMA kernel (0)
    inside one process (rcmarl_wide_set_f16_mode: the library reads RCMARL_WIDE_F16 once).  Restored afterwards."""
    done = []

    def set_form(bk, mode):
        done.append((bk.lib, bk.lib.rcmarl_wide_f16_mode()))
        bk.lib.rcmarl_wide_set_f16_mode(int(mode))
    yield set_form
    for lib, prev in reversed(done):
        lib.rcmarl_wide_set_f16_mode(prev)
