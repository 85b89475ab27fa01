"""Oracle parity at the shapes round 2 left open ("Parity: green, with three loosened bars to tighten"):

   (a) BASELINE configs[4] at FULL width -- 1024 agents, 512-unit critic, circulant d = 66, H = 32, 32x32 grid: one update
       epoch of the whole instance on the GPU, and for a few agents the SAME epoch by the oracle's per-agent methods
       (agents/resilient_CAC_agents.py:103-206 restated in oracle/rpbcac_oracle.py): the 66 local fits of the agent's
       in-neighbourhood, hidden-layer consensus, estimate consensus, projection step -- on the engine's own replay rows.
       (The whole 1024-agent oracle block would sort 1024 x 66 x 1.3 M values on the host; the rows of an agent depend on
       its neighbourhood only.)
   (b) configs[2] at STEADY STATE: 64 agents, random 9-regular + self, H = 4, B = 1000, 2000, 3000, 3000 over four blocks --
       the replay trim, the TD-target row-shift shortcut and the cached activations all live at B = 3000 -- vs oracle.train.
   (c) a deterministic actor bar at 256 agents: after ONE Adam step m = (1 - beta1) g, so the engine's first-moment slots
       against the oracle's give the actor GRADIENT (the parameters themselves can only be held to a statistical bar there:
       Adam turns a gradient of magnitude eps into +-lr): 1e-4 of the largest entry with identical inputs (test_actor_step at
       256 agents, tests/test_kernels_gpu.py), 2e-3 end to end (cancellation amplifies upstream 1e-5 differences).
"""
import numpy as np
import pytest

import engine_checks as EC

pytestmark = pytest.mark.gpu


def test_engine_cfg5_full_shape_rows_vs_oracle():
    worst = EC.check_probed_rows_vs_oracle(1024, 66, 32, 512, 32, "cuda", None, probe=[0, 1023], fast_lr=0.0005)
    print("[parity] cfg5 full shape (1024 agents x 512 units, d=66, H=32): worst |w - w_oracle| / max(1,|w|max) = %.2e over 2 agents "
          "(bar 1e-4)" % worst)


def _random_regular(n, d, seed):
    rng = np.random.default_rng(seed)
    return [[i] + [int(x) for x in rng.permutation([j for j in range(n) if j != i])[:d - 1]] for i in range(n)]


def test_engine_cfg3_steady_state_B3000_vs_oracle():
    n = 64
    args = EC.make_args(["Cooperative"] * n, H=4, n_episodes=200, max_ep_len=20, n_ep_fixed=50, n_epochs=1, buffer_size=2000,
                        seed=64, in_nodes=_random_regular(n, 10, 5), fast_lr=0.005)
    seen = []

    def tweak(eng):
        ub = eng.update_block

        def counted():
            seen.append((eng.B, bool(eng.td_shortcut), bool(eng.rows_episode_aligned)))
            return ub()
        eng.update_block = counted
    W, goals = EC.make_inputs(args, 16, (64,))
    o_logs, o_w = EC.run_oracle(args, 16, 16, "device", (64,), W, goals)
    eng, logs = EC.run_engine(args, 16, 16, "device", "cuda", None, (64,), W, goals, tweak=tweak)
    assert [b for b, _, _ in seen] == [1000, 2000, 3000, 3000], seen          # grow, grow, steady state with the trim live
    assert all(sc and al for _, sc, al in seen), seen                         # shortcut on, rows episode-aligned throughout
    assert eng.lat_active and not eng.k1_circulant
    EC.compare(eng, logs, o_logs, o_w, actor="stat")


def test_actor_gradient_at_256_agents_vs_oracle():
    # round 5: measured median 1.2e-6, worst 4.95e-6 (profiles/r05_parity_worst_cases.txt; round 3's build: 8.3e-4, hence its 2e-3):
    # the bar is the identical-input kernel bar now, 1e-4 (a LeakyReLU knife edge may still move one gradient column: <= 2 % of the agents)
    worst = EC.check_actor_gradient(256, 18, 8, 32, "cuda", None, fast_lr=0.0025, rtol=1e-4)
    print("[parity] actor gradient end to end (Adam m after one step) at 256 agents: worst max|dm| / max|m| = %.2e (bar 1e-4 = the "
          "identical-input kernel bar of test_actor_step)" % worst)
