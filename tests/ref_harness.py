"""Import the reference's own sources from /root/reference under the numpy
`tensorflow`/`gym` stubs of tests/ref_shims (TEST INFRASTRUCTURE).

Only usable in the build container (the GPU box has no /root/reference);
tests that need it are skipped when the directory is absent and rely on the
committed fixtures in tests/golden instead.
"""
import importlib
import os
import sys

REF_ROOT = "/root/reference"
_HERE = os.path.dirname(os.path.abspath(__file__))
_REPO = os.path.dirname(_HERE)


def reference_available():
    return os.path.isdir(os.path.join(REF_ROOT, "agents"))


def load_reference():
    """Returns a namespace with the reference modules (executed verbatim)."""
    if not reference_available():
        raise RuntimeError("reference sources not present")
    for p in (_REPO, os.path.join(_HERE, "ref_shims"), REF_ROOT):
        if p not in sys.path:
            sys.path.insert(0, p) if p != REF_ROOT else sys.path.append(p)
    import tensorflow  # the stub
    assert "ref_shims" in tensorflow.__file__, tensorflow.__file__

    class NS:
        pass

    ns = NS()
    ns.tf = tensorflow
    ns.keras = importlib.import_module("tensorflow.keras")
    ns.resilient = importlib.import_module("agents.resilient_CAC_agents")
    ns.adversarial = importlib.import_module("agents.adversarial_CAC_agents")
    ns.grid_world = importlib.import_module("environments.grid_world")
    ns.train_agents = importlib.import_module("training.train_agents")
    for m in (ns.resilient, ns.adversarial, ns.grid_world, ns.train_agents):
        assert m.__file__.startswith(REF_ROOT), m.__file__
    return ns
