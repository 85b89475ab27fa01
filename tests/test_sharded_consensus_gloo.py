"""C2 (BASELINE configs[4] as ONE instance over several GPUs): agents sharded for phase I, parameter
COLUMNS sharded for the hidden-layer consensus K1, one all-to-all of the message matrix each way.  world_size 2 under
gloo, kernels from the hipemu build: the sharded K1 equals the unsharded K1 BIT FOR BIT, for the general and the
circulant kernel, with a non-cooperative agent among the rows.  CPU-only."""
import os
import socket
import sys

import numpy as np
import pytest
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


def _case(graph, N, d, H, P_hid, ldp, S=2):
    rng = np.random.default_rng(N * 10 + d + (1 if graph == "circ" else 0))
    if graph == "circ":
        nbr = [[(i + k) % N for k in range(d)] for i in range(N)]
    else:
        nbr = [[i] + [int(x) for x in rng.permutation([j for j in range(N) if j != i])[:d - 1]] for i in range(N)]
    coop = np.ones(N, np.int32)
    coop[N - 2] = 0
    base = rng.normal(size=(S, 1, ldp)).astype(np.float32)
    msg = (base + 0.01 * rng.normal(size=(S, N, ldp))).astype(np.float32)
    msg[:, N - 2] = 1e3                         # the adversary's message
    msg[:, :, 7] = msg[:, :1, 7]                # ties
    theta0 = rng.normal(size=(S, N, ldp)).astype(np.float32)
    return nbr, coop, msg, theta0


def _worker(rank, world, port, out_dir, cases):
    sys.path.insert(0, os.path.dirname(_HERE))
    sys.path.insert(0, _HERE)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from emu_util import emu_lib
    from rcmarl_amd.parallel import ShardedConsensus
    lib = emu_lib()
    for ci, (graph, N, d, H, P_hid, ldp) in enumerate(cases):
        nbr, coop, msg, theta0 = _case(graph, N, d, H, P_hid, ldp)
        S = msg.shape[0]
        # unsharded reference: the whole matrix on this rank
        t_msg, t_ref = torch.from_numpy(msg.copy()), torch.from_numpy(theta0.copy())
        t_nbr, t_coop = torch.tensor(np.asarray(nbr, np.int32)), torch.from_numpy(coop.copy())
        lib.rcmarl_consensus_params(t_msg.data_ptr(), t_ref.data_ptr(), t_nbr.data_ptr(), t_coop.data_ptr(), S, N, ldp, P_hid, d,
                                    H, None, None, None)
        # sharded: this rank owns a block of agents; columns are split over the ranks for K1
        sc = ShardedConsensus(lib, S, N, P_hid, d, H, nbr, coop, "cpu")
        assert sc.circulant == (graph == "circ")
        mine = slice(sc.a_lo, sc.a_hi)
        msg_local, theta_local = torch.from_numpy(msg[:, mine].copy()), torch.from_numpy(theta0[:, mine].copy())
        sc.exchange(msg_local)
        np.testing.assert_array_equal(sc.msg_cols[:, :, :sc.width].numpy(), msg[:, :, sc.c_lo:sc.c_hi])     # the transpose itself
        sc.consensus()
        sc.gather(theta_local)
        np.save(os.path.join(out_dir, "c%d_r%d.npy" % (ci, rank)),
                np.stack([theta_local.numpy(), t_ref[:, mine].numpy()]))
    dist.destroy_process_group()


CASES = [("circ", 12, 6, 2, 200, 256), ("rand", 12, 5, 2, 200, 256), ("circ", 9, 4, 1, 50, 64), ("rand", 7, 4, 1, 130, 192)]


def test_column_sharded_k1_equals_unsharded_world2(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), CASES), nprocs=world, join=True)
    for ci in range(len(CASES)):
        for r in range(world):
            got, want = np.load(tmp_path / ("c%d_r%d.npy" % (ci, r)))
            np.testing.assert_array_equal(got, want, err_msg="case %d rank %d" % (ci, r))       # bit for bit


def test_shard_ranges():
    from rcmarl_amd.parallel import agent_range, column_ranges
    assert [agent_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    cr = column_ranges(1311744, 8)
    assert cr[0][0] == 0 and cr[-1][1] == 1311744 and all(a[1] == b[0] for a, b in zip(cr, cr[1:]))
    assert all(c0 % 64 == 0 for c0, _ in cr) and max(c1 - c0 for c0, c1 in cr) - min(c1 - c0 for c0, c1 in cr) <= 64
    assert column_ranges(50, 2) == [(0, 50), (50, 50)]
