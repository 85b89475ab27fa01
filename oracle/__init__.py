"""CPU oracle for the RPBCAC hot path -- TEST INFRASTRUCTURE ONLY.

This package restates, in plain NumPy fp32, the algorithm of
mfigura/Resilient-consensus-based-MARL (agents/resilient_CAC_agents.py,
agents/adversarial_CAC_agents.py, training/train_agents.py,
environments/grid_world.py) plus the Keras/TensorFlow semantics those files
lean on (the reference does not vendor TensorFlow).

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  The product package never does: the HIP path
fails loudly when its extension is missing instead of falling back here.

Parity status: the aggregation rule, the hidden-layer consensus loops, the
whole training loop orchestration and the grid-world are pinned against the
reference *source* executed verbatim under numpy-backed ``tensorflow``/``gym``
stubs (tests/ref_shims, tests/golden).  The arithmetic of TensorFlow itself
(summation order inside its kernels) cannot be run here: "parity unpinned"
with respect to real TensorFlow numerics.
"""
