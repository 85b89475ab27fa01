"""N>1 path on CPU: world_size-2 gloo processes shard seeds and all-reduce the
return curves (the only collective of the path).  CPU-only."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

_HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.dirname(_HERE))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rcmarl_amd.parallel import allreduce_curves, shard_seeds
    seeds = list(range(100, 107))
    mine = shard_seeds(seeds, rank, world)
    E = 6
    curves = np.stack([np.stack([np.sin(s + e) * np.arange(1, 4) for e in range(E)]) for s in mine])   # [n, E, 3]
    mean, std = allreduce_curves(curves.sum(0), len(mine), sq_sums=(curves ** 2).sum(0))
    np.save(os.path.join(out_dir, "r%d.npy" % rank), np.stack([mean, std]))
    dist.destroy_process_group()


def test_seed_sharded_allreduce_world2(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    seeds = list(range(100, 107))
    E = 6
    allc = np.stack([np.stack([np.sin(s + e) * np.arange(1, 4) for e in range(E)]) for s in seeds])
    r0, r1 = np.load(tmp_path / "r0.npy"), np.load(tmp_path / "r1.npy")
    np.testing.assert_array_equal(r0, r1)                       # identical on every rank
    np.testing.assert_allclose(r0[0], allc.mean(0), rtol=1e-12)
    np.testing.assert_allclose(r0[1], allc.std(0), rtol=1e-9, atol=1e-12)


def test_single_process_no_group():
    from rcmarl_amd.parallel import allreduce_curves, shard_seeds
    assert shard_seeds([1, 2, 3, 4, 5], 1, 2) == [2, 4]
    m = allreduce_curves(np.ones((4, 3)) * 6, 3)
    np.testing.assert_allclose(m, 2.0)
