"""Multi-GPU layer: independent seeds are sharded over the ranks (one process per
GPU); the ONLY collective is an all-reduce (sum) of the return-curve
accumulators -- RCCL over xGMI on GPUs (backend "nccl"), gloo in CPU tests.

The reference has no counterpart: it launches one SGE job per seed and averages
the curves offline (plot_results.py:39).  Seeds share nothing, so there is no
data-path collective; the all-reduce is ~tens of KB and latency-bound.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_seeds(all_seeds, rank, world):
    """Round-robin seed -> rank assignment."""
    return [s for k, s in enumerate(all_seeds) if k % world == rank]


def allreduce_curves(local_sums, n_local_seeds, device=None, sq_sums=None):
    """local_sums: [E, C] per-episode sums over this rank's seeds (float64).
    Returns the across-all-seeds mean curve [E, C] (and std if sq_sums given),
    identical on every rank.  Works without an initialised process group (world=1)."""
    local_sums = np.asarray(local_sums, dtype=np.float64)
    parts = [local_sums.ravel(), np.asarray([float(n_local_seeds)])]
    if sq_sums is not None:
        parts.insert(1, np.asarray(sq_sums, dtype=np.float64).ravel())
    buf = torch.from_numpy(np.concatenate(parts))
    if device is not None:
        buf = buf.to(device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    out = buf.cpu().numpy()
    n = out[-1]
    k = local_sums.size
    mean = (out[:k] / n).reshape(local_sums.shape)
    if sq_sums is None:
        return mean
    var = np.maximum(out[k:2 * k].reshape(local_sums.shape) / n - mean ** 2, 0.0)
    return mean, np.sqrt(var)
