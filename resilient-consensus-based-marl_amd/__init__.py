"""rcmarl_amd -- MI355X-native RPBCAC (resilient projection-based consensus
actor-critic) training engine.

Drop-in for the hot path of mfigura/Resilient-consensus-based-MARL:
``agents/resilient_CAC_agents.py``, ``agents/adversarial_CAC_agents.py``,
``training/train_agents.py``, ``environments/grid_world.py`` keep their Python
surface; the per-agent TensorFlow loop underneath is replaced by batched HIP
kernels (csrc/) behind the C-ABI of include/rcmarl.h.
"""
__version__ = "0.1.0"
