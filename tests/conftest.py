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
