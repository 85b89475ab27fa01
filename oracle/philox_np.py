"""Philox4x32-10 in NumPy -- the counter-based stream of the engine's
``rng_mode='device'`` rollout (TEST INFRASTRUCTURE, see oracle/__init__.py).

The reference samples actions from NumPy's global legacy stream
(agents/resilient_CAC_agents.py:208-219); that stream is inherently serial.
The device mode replaces it with a counter-based generator so that every
(seed, episode, step, agent) draw is independent.  This file is the CPU
statement of that generator; csrc/rcmarl_rng.h is the HIP statement.  Both
must produce identical 32-bit words.

counter = (agent, step, episode, stream)   key = (seed_lo, seed_hi)
stream 0: action draws     stream 1: environment reset draws
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = 0x9E3779B9
W1 = 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32(c0, c1, c2, c3, k0, k1, rounds=10):
    """All arguments broadcastable uint32 arrays/ints.  Returns 4 uint32 arrays."""
    c0, c1, c2, c3 = [np.asarray(c, dtype=np.uint64) & MASK for c in np.broadcast_arrays(c0, c1, c2, c3)]
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for r in range(rounds):
        p0 = M0 * c0
        p1 = M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0)), lo1, (hi0 ^ c3 ^ np.uint64(k1)), lo0
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def mulhi_range(r, n):
    """Map a uint32 word to [0, n) with the multiply-high trick."""
    return ((np.asarray(r, dtype=np.uint64) * np.uint64(n)) >> np.uint64(32)).astype(np.int64)


def u01(r):
    """Top 24 bits -> float32 in [0,1)."""
    return ((np.asarray(r, dtype=np.uint32) >> np.uint32(8)).astype(np.float32)
            * np.float32(1.0 / 16777216.0))


def seed_key(seed):
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    return seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF


def sample_actions(probs, seed, episode, step, mu=0.1):
    """Device-mode action draw for all agents of one seed.
    probs: [N, A] float32 actor outputs.  Mirrors csrc `rcmarl_sample_action`:
    a_rand = mulhi(r0, A); u1,u2 = u01(r1), u01(r2);
    a_pol = #{k < A-1 : u1 >= cumsum_fp32(p)[k]};  action = a_pol if u2 < 1-mu else a_rand."""
    probs = np.asarray(probs, dtype=np.float32)
    N, A = probs.shape
    k0, k1 = seed_key(seed)
    r0, r1, r2, _ = philox4x32(np.arange(N), step, episode, 0, k0, k1)
    a_rand = mulhi_range(r0, A)
    u1, u2 = u01(r1), u01(r2)
    c = np.zeros(N, np.float32)
    a_pol = np.zeros(N, np.int64)
    for k in range(A - 1):
        c = (c + probs[:, k]).astype(np.float32)
        a_pol += (u1 >= c)
    thr = np.float32(1.0) - np.float32(mu)
    return np.where(u2 < thr, a_pol, a_rand).astype(np.int64)


def reset_positions(n_agents, nrow, ncol, seed, episode):
    """Device-mode replacement of np.random.randint([0,0],[nrow,ncol],(N,2))
    (environments/grid_world.py:40)."""
    k0, k1 = seed_key(seed)
    r0, r1, _, _ = philox4x32(np.arange(n_agents), 0, episode, 1, k0, k1)
    return np.stack([mulhi_range(r0, nrow), mulhi_range(r1, ncol)], axis=1)
