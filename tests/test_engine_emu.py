"""End-to-end engine (host logic + every kernel, hipemu build) vs the oracle's
train() on tiny cooperative scenarios, both RNG modes.  CPU-only."""
import pytest

import engine_checks as EC
from emu_util import emu_lib


@pytest.mark.parametrize("rng_mode", ["device", "numpy"])
def test_engine_coop_matches_oracle(rng_mode):
    args = EC.make_args(["Cooperative"] * 5, H=1, n_episodes=5, max_ep_len=3, n_ep_fixed=2, n_epochs=1, buffer_size=9, seed=11)
    eng, logs, o_logs, o_w = EC.run_pair(args, 5, 5, rng_mode, "cpu", emu_lib(), seeds=(11, 12))
    EC.compare(eng, logs, o_logs, o_w)


def test_engine_common_reward_H0():
    args = EC.make_args(["Cooperative"] * 5, H=0, n_episodes=4, max_ep_len=3, n_ep_fixed=2, n_epochs=1, buffer_size=9, seed=5,
                        common_reward=True)
    eng, logs, o_logs, o_w = EC.run_pair(args, 6, 6, "device", "cpu", emu_lib(), seeds=(5,))
    EC.compare(eng, logs, o_logs, o_w)


@pytest.mark.parametrize("labels", [["Cooperative"] * 4 + ["Malicious"], ["Cooperative", "Greedy", "Cooperative", "Cooperative", "Faulty"]])
def test_engine_with_adversaries(labels):
    """BASELINE configs[1]: 4 cooperative + 1 adversary, H=1 (mini-batch message generators in one launch per fit)."""
    args = EC.make_args(labels, H=1, n_episodes=4, max_ep_len=4, n_ep_fixed=2, n_epochs=2, buffer_size=12, seed=21)
    eng, logs, o_logs, o_w = EC.run_pair(args, 5, 5, "device", "cpu", emu_lib(), seeds=(21, 22))
    EC.compare(eng, logs, o_logs, o_w)


@pytest.mark.parametrize("labels,nrow,ncol", [(["Cooperative"] * 5, 5, 5), (["Cooperative"] * 4 + ["Malicious"], 7, 4)])
def test_engine_lattice_path_matches_oracle(labels, nrow, ncol):
    """Same end-to-end check with the layer-1 GEMMs forced onto the exact bf16x3 (lattice) kernels."""
    args = EC.make_args(labels, H=1, n_episodes=4, max_ep_len=3, n_ep_fixed=2, n_epochs=2, buffer_size=9, seed=31)
    eng, logs, o_logs, o_w = EC.run_pair(args, nrow, ncol, "device", "cpu", emu_lib(), seeds=(31, 32), lattice=True)
    assert eng.lat_enabled and eng.lat_active          # the replay rows passed the lattice check -> path was used
    EC.compare(eng, logs, o_logs, o_w)


@pytest.mark.parametrize("rng_mode,critic_hid,H,lattice", [("numpy", 24, 0, False), ("device", 40, 1, True)])
def test_engine_wide_critic_matches_oracle(rng_mode, critic_hid, H, lattice):
    """BASELINE configs[4] in miniature: a critic wider than the reference's 20 units runs the dense-GEMM path
    (csrc/wide_kernels.hip) for its local fits, TD targets, estimate consensus and start-state values."""
    args = EC.make_args(["Cooperative"] * 5, H=H, n_episodes=4, max_ep_len=3, n_ep_fixed=2, n_epochs=1, buffer_size=9, seed=41)
    eng, logs, o_logs, o_w = EC.run_pair(args, 5, 5, rng_mode, "cpu", emu_lib(), seeds=(41,), critic_hid=critic_hid,
                                         lattice=lattice)
    assert eng.wide and eng.lat_active == lattice      # lattice: layer 1 of the wide critic on the bf16x3 kernels
    EC.compare(eng, logs, o_logs, o_w)


@pytest.mark.parametrize("labels,rng_mode", [(["Cooperative"] * 4 + ["Greedy"], "device"), (["Cooperative"] * 5, "numpy")])
def test_checkpoint_resume_is_bit_identical(labels, rng_mode, tmp_path):
    """Train 3 blocks straight vs 1 block -> save -> fresh engine -> load -> 2 blocks: same logs, same bits
    (weights, Adam slots, replay rows, RNG position all travel in the checkpoint)."""
    import numpy as np
    from rcmarl_amd.engine import EngineConfig, RPBCACEngine

    def make():
        cfg = EngineConfig(5, labels, EC.CIRC5, H=1, max_ep_len=3, n_ep_fixed=2, n_epochs=1, buffer_size=9, nrow=5, ncol=5,
                           n_seeds=2, rng_mode=rng_mode, lattice=True)
        eng = RPBCACEngine(cfg, seeds=[7, 8], device="cpu", lib=emu_lib())
        eng.init_glorot(base_seed=3)
        eng.set_goals(np.array([[1, 2], [0, 0], [4, 4], [2, 3], [3, 1]]))
        if rng_mode == "numpy":
            eng.np_rngs = [np.random.RandomState(70 + s) for s in range(2)]
        return eng
    a = make()
    la = a.train(6)
    b = make()
    lb1 = b.train(2)
    b.save_checkpoint(str(tmp_path / "ck.pt"))
    c = make()
    c.init_glorot(base_seed=99)                       # different weights: everything must come from the file
    c.load_checkpoint(str(tmp_path / "ck.pt"))
    lc = c.train(4)
    for k in la:
        np.testing.assert_array_equal(la[k], np.concatenate([lb1[k], lc[k]], axis=0))
    for net in ("actor", "critic", "tr"):
        np.testing.assert_array_equal(a.get_all_weights(net), c.get_all_weights(net))
    np.testing.assert_array_equal(a.adam_m.numpy(), c.adam_m.numpy())


def test_engine_buffer_not_a_multiple_of_the_episode_length():
    """The replay trim may cut an episode in two (buffer_size % max_ep_len != 0): the TD-target shortcut that reads the
    next state's value from the following row must switch itself off; results still match the oracle."""
    args = EC.make_args(["Cooperative"] * 5, H=1, n_episodes=6, max_ep_len=3, n_ep_fixed=2, n_epochs=2, buffer_size=7, seed=61)
    eng, logs, o_logs, o_w = EC.run_pair(args, 5, 5, "device", "cpu", emu_lib(), seeds=(61,))
    assert not eng.rows_episode_aligned
    EC.compare(eng, logs, o_logs, o_w)
