"""Independent cross-checks of the oracle's Keras-semantics restatement:
(1) torch-CPU autograd (a different implementation of the same math),
(2) the committed fixtures regenerate identically from the live reference
    sources when /root/reference is present (build container only).
CPU-only."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import mlp_np as M
import ref_harness


def _torch_forward(params, x):
    W1, b1, W2, b2, W3, b3 = params
    a1 = torch.nn.functional.leaky_relu(x @ W1 + b1, 0.1)
    a2 = torch.nn.functional.leaky_relu(a1 @ W2 + b2, 0.1)
    return a2 @ W3 + b3


def test_mse_sgd_step_vs_torch():
    rng = np.random.default_rng(0)
    p = M.init_mlp(rng, 10, 20, 1)
    for k in (1, 3, 5):
        p[k] += rng.normal(size=p[k].shape).astype(np.float32) * 0.1
    x = rng.normal(size=(64, 10)).astype(np.float32)
    y = rng.normal(size=(64, 1)).astype(np.float32)
    w = rng.uniform(0.1, 2, size=64).astype(np.float32)
    tp = [torch.tensor(a, requires_grad=True) for a in p]
    pred = _torch_forward(tp, torch.tensor(x))
    loss_t = (((pred - torch.tensor(y)) ** 2).mean(-1) * torch.tensor(w)).sum() / 64
    loss_t.backward()
    out, cache = M.forward(p, x, want_cache=True)
    loss, dout = M.mse_loss_and_dout(out, y, sample_weight=w)
    grads = M.backward(p, cache, dout)
    assert abs(float(loss) - float(loss_t.detach())) < 1e-6 * max(1, abs(float(loss_t.detach())))
    for g, t in zip(grads, tp):
        np.testing.assert_allclose(g, t.grad.numpy(), rtol=2e-5, atol=1e-7)


def test_sparse_ce_adam_vs_torch():
    """Loss/gradients vs torch autograd at every step; the Adam update vs the
    float64 closed form of the TF2 ResourceApplyAdam recurrence."""
    rng = np.random.default_rng(1)
    p = M.init_mlp(rng, 10, 20, 5)
    x = rng.normal(size=(50, 10)).astype(np.float32)
    lab = rng.integers(0, 5, size=50)
    w = rng.normal(size=50).astype(np.float32)
    st = M.AdamState(p, 0.002)
    m64 = [np.zeros(a.shape) for a in p]
    v64 = [np.zeros(a.shape) for a in p]
    for step in range(1, 4):
        tp = [torch.tensor(a, requires_grad=True) for a in p]
        logits = _torch_forward(tp, torch.tensor(x))
        ce = torch.nn.functional.cross_entropy(logits, torch.tensor(lab), reduction="none")
        loss_t = (ce * torch.tensor(w)).sum() / 50
        loss_t.backward()
        lg, cache = M.forward(p, x, want_cache=True)
        loss, dl = M.sparse_ce_loss_and_dlogits(lg, lab, w)
        grads = M.backward(p, cache, dl)
        assert abs(float(loss) - float(loss_t.detach())) < 2e-6 * max(1, abs(float(loss_t.detach())))
        for g, t in zip(grads, tp):
            np.testing.assert_allclose(g, t.grad.numpy(), rtol=2e-5, atol=1e-7)
        before = [a.astype(np.float64) for a in p]
        M.adam_apply(p, grads, st)
        alpha = 0.002 * np.sqrt(1 - 0.999 ** step) / (1 - 0.9 ** step)
        for k, g in enumerate(grads):
            g = g.astype(np.float64)
            m64[k] += (g - m64[k]) * 0.1
            v64[k] += (g * g - v64[k]) * 0.001
            want = before[k] - alpha * m64[k] / (np.sqrt(v64[k]) + 1e-7)
            np.testing.assert_allclose(p[k], want, rtol=1e-5, atol=1e-7)
            m64[k], v64[k] = st.m[k].astype(np.float64), st.v[k].astype(np.float64)


@pytest.mark.reference
@pytest.mark.skipif(not ref_harness.reference_available(), reason="needs /root/reference (build container)")
def test_fixtures_regenerate_from_live_reference(tmp_path):
    """Re-run tests/golden/make_golden.py against the live reference sources and
    compare every array with the committed fixture."""
    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys, numpy as np; sys.argv=['x']; sys.path.insert(0, %r); import runpy;"
            "ns = runpy.run_path(%r, run_name='gen');"
            "out = {}; [ns[f](out) for f in ('gen_aggregation','gen_hidden_consensus_fixture','gen_shipped_artifacts','gen_env','gen_training')];"
            "np.savez(%r, **out)") % (here, os.path.join(here, "golden", "make_golden.py"), str(tmp_path / "g.npz"))
    subprocess.run([sys.executable, "-c", code], check=True, capture_output=True, cwd=os.path.dirname(here))
    new = np.load(tmp_path / "g.npz")
    old = np.load(os.path.join(here, "golden", "reference_golden.npz"))
    assert sorted(new.files) == sorted(old.files)
    for k in old.files:
        np.testing.assert_array_equal(new[k], old[k], err_msg=k)
