"""Diagnostic (GPU): per-parameter errors of one plain SGD fit (tests/kernel_checks.check_sgd_fit) under each mid kernel form."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import kernel_checks as KC
from test_kernels_gpu import GpuBackend
bk = GpuBackend()
errs = []
def rc(got, want, rtol, what=""):
    scale = max(1.0, float(np.abs(want).max()))
    errs.append((what, float(np.abs(got - want).max()) / scale))
KC.rel_close = rc
for shape in ((1, 128, 333, 256), (1, 128, 256, 256), (1, 128, 1000, 256), (1, 16, 333, 256), (1, 128, 333, 64)):
    for kind in ("5", "7", "8"):
        os.environ["RCMARL_MIDFIT"] = kind
        errs.clear()
        try:
            KC.check_sgd_fit(bk, *shape, steps=5, masked_agent=None)
        except AssertionError as e:
            errs.append(("assert " + str(e)[:60], 0.0))
        d = collections.defaultdict(float)
        for w, e in errs:
            d[w] = max(d[w], e)
        print(shape, "v" + kind, {k.replace("fit param ", "p"): "%.1e" % v for k, v in d.items()}, flush=True)
