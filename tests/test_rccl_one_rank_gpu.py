"""Every collective of the multi-GPU layer through RCCL on the ONE GPU a test box has.

`torch.distributed` backend "nccl" IS RCCL on ROCm.  A one-rank group is the only RCCL group a single-GPU box can build, and
it is enough to execute what a world-size-2 gloo test cannot: `init_process_group("nccl", device_id=...)`, an all-reduce of
the return-curve accumulator on a DEVICE tensor (C1, parallel.allreduce_curves), `all_to_all_single` with explicit split
sizes on views of the message matrix and `all_gather` on device rows (C2, parallel.ShardedConsensus / RPBCACEngine.
shard_agents(force=True)).  The sharded results must equal the unsharded kernels' BIT FOR BIT.  Replaces the reference's
in-process gather `[critic_weights[i] for i in in_nodes[node]]` (training/train_agents.py:129-130) and its one-SGE-job-per-
seed launch (simulation_results/raw_data/coop/H=1/seed=100/job.sh:4).  Runs in a subprocess: the process group must not
leak into the other GPU tests."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
_HERE = os.path.dirname(os.path.abspath(__file__))

_SCRIPT = r'''
import json, os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
out = {"backend": dist.get_backend(), "world": dist.get_world_size(), "rccl": ".".join(str(v) for v in torch.cuda.nccl.version())}
from rcmarl_amd import capi
from rcmarl_amd.parallel import ShardedConsensus, TorchComm, allreduce_curves
lib = capi.load()

# C1: the return-curve all-reduce on a device buffer
sums = np.arange(12, dtype=np.float64).reshape(4, 3)
mean, std = allreduce_curves(sums, 2, device=dev, sq_sums=sums ** 2)
assert np.array_equal(mean, sums / 2), mean
out["allreduce_curves"] = "ok"

# C2: all-to-all transposes around K1, general and circulant kernel, one and two seeds, a non-cooperative row
rng = np.random.default_rng(3)
for graph, S, N, d, H, P_hid, ldp in (("circ", 1, 12, 6, 2, 200, 256), ("rand", 1, 12, 5, 2, 200, 256), ("circ", 2, 9, 4, 1, 130, 192)):
    nbr = [[(i + k) % N for k in range(d)] for i in range(N)] if graph == "circ" else \
          [[i] + [int(x) for x in rng.permutation([j for j in range(N) if j != i])[:d - 1]] for i in range(N)]
    coop = np.ones(N, np.int32); coop[N - 2] = 0
    msg = (rng.normal(size=(S, 1, ldp)) + 0.01 * rng.normal(size=(S, N, ldp))).astype(np.float32); msg[:, N - 2] = 1e3
    theta0 = rng.normal(size=(S, N, ldp)).astype(np.float32)
    t_msg, t_ref = torch.from_numpy(msg).to(dev), torch.from_numpy(theta0.copy()).to(dev)
    t_nbr, t_coop = torch.tensor(np.asarray(nbr, np.int32), device=dev), torch.from_numpy(coop).to(dev)
    lib.rcmarl_consensus_params(t_msg.data_ptr(), t_ref.data_ptr(), t_nbr.data_ptr(), t_coop.data_ptr(), S, N, ldp, P_hid, d, H,
                                None, None, None)
    sc = ShardedConsensus(lib, S, N, P_hid, d, H, nbr, coop, dev, comm=TorchComm(), force_collectives=True)
    theta = torch.from_numpy(theta0.copy()).to(dev)
    sc.exchange(t_msg); sc.consensus(); sc.gather(theta)
    torch.cuda.synchronize()
    assert np.array_equal(sc.msg_cols[:, :, :sc.width].cpu().numpy(), msg[:, :, :P_hid])
    assert np.array_equal(theta.cpu().numpy(), t_ref.cpu().numpy()), (graph, S)
    assert sc.passes == (2 if S == 1 else 4), sc.passes          # one pack pass per direction (+ one staging pass for S > 1)
out["sharded_consensus"] = "ok"

# the agent-sharded wide-critic engine with every collective over the one-rank RCCL group == the unsharded engine
import engine_checks as EC
n, d, H, hid = 8, 4, 1, 64
nodes = [[(i + k) % n for k in range(d)] for i in range(n)]
args = EC.make_args(["Cooperative"] * n, H=H, n_episodes=9, max_ep_len=5, n_ep_fixed=4, n_epochs=2, buffer_size=30, seed=23,
                    in_nodes=nodes, fast_lr=0.002)
W, goals = EC.make_inputs(args, 6, (23,), critic_hid=hid)
def snap(eng, logs):
    o = {"theta_" + k: v.cpu().numpy() for k, v in eng.theta.items()}
    o.update({"log_" + k: np.asarray(v) for k, v in logs.items()})
    return o
ref_eng, ref_logs = EC.run_engine(args, 6, 6, "device", "cuda", lib, (23,), W, goals, lattice=True, critic_hid=hid)
counts = {"a2a": 0, "gather": 0}
def tweak(e):
    e.shard_agents(force=True)
    assert e.shard is not None and e.shard.world == 1 and isinstance(e.shard.comm, TorchComm)
    a2a, ag = e.shard.comm.all_to_all_single, e.shard.comm.all_gather
    def c_a2a(*a, **k):
        counts["a2a"] += 1; return a2a(*a, **k)
    def c_ag(*a, **k):
        counts["gather"] += 1; return ag(*a, **k)
    e.shard.comm.all_to_all_single, e.shard.comm.all_gather = c_a2a, c_ag
eng, logs = EC.run_engine(args, 6, 6, "device", "cuda", lib, (23,), W, goals, lattice=True, critic_hid=hid, tweak=tweak)
a, b = snap(ref_eng, ref_logs), snap(eng, logs)
for k in a:
    assert np.array_equal(a[k], b[k]), k
assert counts["a2a"] >= 2 * 2 * 2 and counts["gather"] > 0, counts       # 2 blocks x 2 epochs x (exchange + gather) for the critic
out["sharded_engine"] = counts
dist.barrier()
dist.destroy_process_group()
print("RESULT " + json.dumps(out))
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_every_collective_runs_under_rccl_with_one_rank():
    env = dict(os.environ)
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", "HERE = %r\n" % _HERE + _SCRIPT], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + "\n" + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    out = json.loads(line[len("RESULT "):])
    assert out["backend"] == "nccl" and out["world"] == 1 and out["rccl"]
    assert out["allreduce_curves"] == "ok" and out["sharded_consensus"] == "ok" and out["sharded_engine"]["a2a"] > 0
    os.makedirs(os.path.join(os.path.dirname(_HERE), "gpurun_out"), exist_ok=True)
    with open(os.path.join(os.path.dirname(_HERE), "gpurun_out", "rccl_one_rank.json"), "w") as f:
        json.dump(out, f)
