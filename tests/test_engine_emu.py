"""End-to-end engine (host logic + every kernel, hipemu build) vs the oracle's
train() on tiny cooperative scenarios, both RNG modes.  CPU-only."""
import numpy as np
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


@pytest.mark.parametrize("labels", [["Cooperative"] * 5, ["Cooperative", "Faulty", "Cooperative", "Cooperative", "Malicious"]])
def test_engine_local_fits_as_matrix_core_chains(labels, monkeypatch):
    """RCMARL_FIT_CHAINS=1: the cooperative agents' 5-step full-batch local fits of the small networks through rcmarl_minibatch_fit
    (batch_size = B, no shuffle: one matrix-core wavefront per network; the default from 1024 networks per launch on) vs the oracle."""
    monkeypatch.setenv("RCMARL_FIT_CHAINS", "1")
    args = EC.make_args(labels, H=1, n_episodes=6, max_ep_len=4, n_ep_fixed=2, n_epochs=2, buffer_size=16, seed=31)
    eng, logs, o_logs, o_w = EC.run_pair(args, 5, 5, "device", "cpu", emu_lib(), seeds=(31, 32))
    assert eng._fit_as_chains("critic", eng.coop) and eng._fit_as_chains("tr", eng.coop)
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


def test_engine_wide_critic_on_packed_operands_matches_oracle():
    """A critic width that is a multiple of 128 beside the lattice layer 1: every dense layer of its local fits, TD targets and
    the estimate consensus runs on pre-split packed operands (csrc/dense_pk.hip; RPBCACEngine._local_fit_wide_pk)."""
    args = EC.make_args(["Cooperative"] * 5, H=1, n_episodes=4, max_ep_len=3, n_ep_fixed=2, n_epochs=2, buffer_size=9, seed=41)
    eng, logs, o_logs, o_w = EC.run_pair(args, 5, 5, "device", "cpu", emu_lib(), seeds=(41,), critic_hid=128, lattice=True)
    assert eng.wide and eng.lat_active and eng.pk is not None and eng._pk_ok("critic", "s", eng.lat_B)
    EC.compare(eng, logs, o_logs, o_w)


def test_update_block_from_injected_state_matches_oracle():
    """One update block from IDENTICAL state (tests/engine_checks.check_block_from_injected_state): after two blocks and the rollout
    of the third the engine's weights, Adam slots and replay rows go into oracle.update_block (training/train_agents.py:100-153);
    the third block then runs on both sides at the steady-state batch B = buffer_size + n_ep_fixed * max_ep_len with live actors."""
    args = EC.make_args(["Cooperative"] * 5, H=1, n_episodes=0, max_ep_len=3, n_ep_fixed=2, n_epochs=2, buffer_size=12, seed=71)
    eng, errs, o_w = EC.check_block_from_injected_state(args, 5, 5, "cpu", emu_lib(), (71, 72))
    assert eng.B == 12 and eng.adam_t == 3                       # (the trim after the update brought the 18 rows back to buffer_size)
    assert max(float(e.max()) for e in errs.values()) <= 1e-5, errs
    worst = max(float(np.abs(a - b).max()) for s in range(2) for i in range(5) for a, b in zip(eng.get_weights(s, i, "actor"), o_w[s][i][0]))
    assert worst <= 1e-5, worst


def test_oracle_block_spread_over_processes_equals_the_serial_oracle():
    """tests/engine_checks.run_oracle_blocks_parallel calls the oracle's per-agent methods from a process pool, epoch by epoch (the
    256-agent blocks of the GPU suite take minutes in one process): same weights as oracle.update_block's serial loop, bit for bit."""
    args = EC.make_args(["Cooperative"] * 5, H=1, n_episodes=0, max_ep_len=3, n_ep_fixed=2, n_epochs=2, buffer_size=12, seed=71)
    eng, snaps = EC.check_block_from_injected_state(args, 5, 5, "cpu", emu_lib(), (71, 72), oracle_later=True)
    a = EC.run_oracle_blocks_parallel(dict(args), snaps)
    b = EC.run_oracle_blocks_parallel(dict(args), snaps, force_serial=True)
    for wa, wb in zip(a, b):
        for ag_a, ag_b in zip(wa, wb):
            for net_a, net_b in zip(ag_a, ag_b):
                for x, y in zip(net_a, net_b):
                    np.testing.assert_array_equal(x, y)


def test_engine_wide_critic_with_faulty_agent_matches_oracle():
    """A Faulty agent (frozen critic / team-reward net, learning actor: adversarial_CAC_agents.py:5-55) beside a wide critic:
    its frozen wide message enters every neighbour's aggregation, its actor takes the mini-batch Adam steps from TD errors
    of its own (wide) critic."""
    args = EC.make_args(["Cooperative"] * 4 + ["Faulty"], H=1, n_episodes=4, max_ep_len=3, n_ep_fixed=2, n_epochs=1, buffer_size=9,
                        seed=43)
    eng, logs, o_logs, o_w = EC.run_pair(args, 5, 5, "device", "cpu", emu_lib(), seeds=(43,), critic_hid=24, lattice=False)
    assert eng.wide and hasattr(eng, "adv")
    EC.compare(eng, logs, o_logs, o_w)


@pytest.mark.parametrize("labels", [["Cooperative"] * 3 + ["Greedy", "Malicious"], ["Malicious"] + ["Cooperative"] * 4])
def test_engine_wide_critic_with_greedy_and_malicious_agents_matches_oracle(labels):
    """Greedy / Malicious agents beside a WIDE critic (agents/adversarial_CAC_agents.py:121-165,228-253 take any Keras model):
    their fit(batch_size=32, epochs=10) message generators run through the dense per-agent GEMM entry points, one mini-batch
    at a time (engine_adversaries._fit_critic_family); the Malicious agent's private wide critic too."""
    args = EC.make_args(labels, H=1, n_episodes=4, max_ep_len=3, n_ep_fixed=2, n_epochs=1, buffer_size=9, seed=44)
    eng, logs, o_logs, o_w = EC.run_pair(args, 5, 5, "device", "cpu", emu_lib(), seeds=(44,), critic_hid=24, lattice=False)
    assert eng.wide and hasattr(eng, "adv") and eng.adv.fit
    EC.compare(eng, logs, o_logs, o_w)


@pytest.mark.parametrize("labels,rng_mode", [(["Cooperative"] * 4 + ["Greedy"], "device"), (["Cooperative"] * 5, "numpy")])
def test_checkpoint_resume_is_bit_identical(labels, rng_mode, tmp_path):
    # (the emulated lattice GEMMs are slow and have their own tests: one of the two runs stays off them; (1, 2) blocks on the GPU)
    EC.check_checkpoint_resume(labels, rng_mode, "cpu", emu_lib(), str(tmp_path / "ck.pt"), blocks=(1, 1), lattice=rng_mode == "device")


def test_engine_buffer_not_a_multiple_of_the_episode_length():
    """The replay trim may cut an episode in two (buffer_size % max_ep_len != 0): the TD-target shortcut that reads the
    next state's value from the following row must switch itself off; results still match the oracle."""
    args = EC.make_args(["Cooperative"] * 5, H=1, n_episodes=6, max_ep_len=3, n_ep_fixed=2, n_epochs=2, buffer_size=7, seed=61)
    eng, logs, o_logs, o_w = EC.run_pair(args, 5, 5, "device", "cpu", emu_lib(), seeds=(61,))
    assert not eng.rows_episode_aligned
    EC.compare(eng, logs, o_logs, o_w)


def test_replay_grows_past_the_steady_state_capacity():
    """The reference's replay lists have no capacity (training/train_agents.py:76-80; the trim to buffer_size happens after
    an update, :158-163): trailing episodes of one train() call stay in the buffer, and a second call on the same state
    pushes B past buffer_size + n_ep_fixed*max_ep_len.  The engine grows its row-sized buffers; results follow the oracle
    run with the same carried-over exp_buffer."""
    import numpy as np
    from oracle import rpbcac_oracle as O
    args = EC.make_args(["Cooperative"] * 5, H=1, n_episodes=5, max_ep_len=3, n_ep_fixed=2, n_epochs=1, buffer_size=9, seed=71)
    seeds = (71,)
    W, goals = EC.make_inputs(args, 5, seeds)
    # oracle: two calls, the replay lists and the agents carried over
    a1, a2 = dict(args), dict(args, n_episodes=4)
    agents = [O.make_agent("Cooperative", W[0][i]["actor"], W[0][i]["critic"], W[0][i]["tr"], 0.002, 0.01, 0.9, 1) for i in range(5)]
    np.random.seed(71)
    env = O.GridWorldOracle(5, 5, 5, goals[0], None, True, True)
    buf = ([], [], [], [])
    _, df1 = O.train(env, agents, a1, exp_buffer=buf, rng_mode="numpy")
    rows_after_first = len(buf[0])
    ow, df2 = O.train(env, agents, a2, exp_buffer=buf, rng_mode="numpy")
    # engine: two train() calls on one engine
    caps = []

    def tweak(eng):
        caps.append(eng.cap)
    eng, l1 = EC.run_engine(args, 5, 5, "numpy", "cpu", emu_lib(), seeds, W, goals, tweak=tweak)
    assert eng.B == rows_after_first and eng.B > args["buffer_size"]            # trailing episode: untrimmed rows
    l2 = eng.train(4)
    assert eng.cap > caps[0]                                                        # 12 + 6 rows > 9 + 6
    logs = {k: np.concatenate([l1[k], l2[k]], axis=0) for k in l1}
    import pandas as pd
    EC.compare(eng, logs, [pd.concat([df1, df2], ignore_index=True)], [ow])



def test_probed_rows_vs_oracle_wide_critic_small():
    """tests/test_engine_round3_parity_gpu.py::test_engine_cfg5_full_shape_rows_vs_oracle in miniature (hipemu kernels):
    one epoch of a wide-critic instance vs the oracle's per-agent methods for two probed agents."""
    worst = EC.check_probed_rows_vs_oracle(8, 4, 1, 32, 5, "cpu", emu_lib(), probe=[0, 7], fast_lr=0.01, n_ep_fixed=2, max_ep_len=5)
    assert worst <= 1e-4


def test_actor_gradient_vs_oracle_small():
    """...::test_actor_gradient_at_256_agents_vs_oracle in miniature: Adam m after one step == (1 - beta1) x oracle gradient."""
    worst = EC.check_actor_gradient(6, 4, 1, 5, "cpu", emu_lib(), fast_lr=0.01, n_ep_fixed=2, max_ep_len=5)
    assert worst <= 1e-4
