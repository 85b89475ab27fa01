"""End-to-end parity on a real MI355X: the batched engine vs the oracle's train()
from identical initial weights and identical RNG streams."""
import pytest

import engine_checks as EC

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rng_mode", ["device", "numpy"])
def test_engine_coop_H1(rng_mode):
    args = EC.make_args(["Cooperative"] * 5, H=1, n_episodes=45, max_ep_len=20, n_ep_fixed=10, n_epochs=3, buffer_size=400, seed=100)
    eng, logs, o_logs, o_w = EC.run_pair(args, 5, 5, rng_mode, "cuda", None, seeds=(100, 200, 300))
    EC.compare(eng, logs, o_logs, o_w)


def test_engine_coop_H0_common_reward():
    args = EC.make_args(["Cooperative"] * 5, H=0, n_episodes=20, max_ep_len=20, n_ep_fixed=10, n_epochs=2, buffer_size=300, seed=7,
                        common_reward=True)
    eng, logs, o_logs, o_w = EC.run_pair(args, 5, 5, "device", "cuda", None, seeds=(7,))
    EC.compare(eng, logs, o_logs, o_w)


def test_engine_12_agents_H2():
    n = 12
    in_nodes = [[(i + k) % n for k in range(6)] for i in range(n)]
    args = EC.make_args(["Cooperative"] * n, H=2, n_episodes=10, max_ep_len=10, n_ep_fixed=5, n_epochs=2, buffer_size=80, seed=9,
                        in_nodes=in_nodes)
    eng, logs, o_logs, o_w = EC.run_pair(args, 8, 8, "device", "cuda", None, seeds=(9, 10))
    EC.compare(eng, logs, o_logs, o_w)


@pytest.mark.parametrize("labels", [["Cooperative"] * 4 + ["Malicious"], ["Cooperative", "Greedy", "Cooperative", "Cooperative", "Faulty"]])
def test_engine_with_adversaries(labels):
    """BASELINE configs[1] (4 cooperative + 1 Malicious, H=1) and a Greedy+Faulty mix: shuffled
    mini-batch fits (32 x 10 epochs) and shuffled actor fits (200-row Adam mini-batches)."""
    args = EC.make_args(labels, H=1, n_episodes=30, max_ep_len=20, n_ep_fixed=15, n_epochs=2, buffer_size=450, seed=300)
    eng, logs, o_logs, o_w = EC.run_pair(args, 5, 5, "device", "cuda", None, seeds=(300, 301))
    EC.compare(eng, logs, o_logs, o_w, rtol_w=5e-4)


@pytest.mark.parametrize("n,nrow,ncol,labels", [(5, 5, 5, None), (20, 16, 12, None), (5, 5, 5, ["Cooperative"] * 4 + ["Malicious"])])
def test_engine_lattice_path(n, nrow, ncol, labels):
    """Layer-1 GEMMs on the exact bf16x3 (lattice) kernels: same parity bar as the f32-MFMA path.
    n=20 is above the engine's own "auto" threshold; n=5 forces the path on the reference's config."""
    labels = labels or ["Cooperative"] * n
    d = 4 if n == 5 else 6
    in_nodes = [[(i + k) % n for k in range(d)] for i in range(n)]
    args = EC.make_args(labels, H=1 if n == 5 else 2, n_episodes=20, max_ep_len=20, n_ep_fixed=10, n_epochs=3, buffer_size=300,
                        seed=41, in_nodes=in_nodes)
    eng, logs, o_logs, o_w = EC.run_pair(args, nrow, ncol, "device", "cuda", None, seeds=(41, 42), lattice=True if n == 5 else "auto")
    assert eng.lat_enabled and eng.lat_active
    EC.compare(eng, logs, o_logs, o_w, rtol_w=5e-4 if "Malicious" in labels else 2e-4)
