"""Kernel-vs-oracle parity on the hipemu (CPU) build of the HIP sources.
Exercises indexing / LDS staging / barrier / MFMA-layout logic without a GPU.
The real parity tests (same checks, real hardware, product C-ABI) are in
test_kernels_gpu.py."""
import numpy as np
import pytest

import kernel_checks as KC
from emu_util import emu_lib, p


class EmuBackend:
    stream = None

    def __init__(self):
        self.lib = emu_lib()

    def dev(self, arr):
        return np.ascontiguousarray(arr).copy()

    def ptr(self, h):
        return p(h)

    def host(self, h):
        return h


@pytest.fixture(scope="module")
def bk():
    return EmuBackend()


@pytest.mark.parametrize("N,d,H,P,P_hid,graph", [
    (5, 4, 1, 661, 640, "circ"), (5, 4, 0, 661, 640, "circ"), (12, 10, 4, 200, 150, "rand"),
    (7, 3, 1, 70, 64, "circ"), (20, 18, 8, 130, 128, "circ"), (9, 7, 2, 100, 90, "rand"),
    (30, 23, 5, 90, 77, "rand"),   # no generated network for (23,5): rank-counting fallback
])
def test_consensus_params(bk, N, d, H, P, P_hid, graph):
    KC.check_consensus_params(bk, N, d, H, P, P_hid, graph)


@pytest.mark.parametrize("N,d,H,P,P_hid", [(5, 4, 1, 150, 130),       # G = 2, wraps in every group but the first
                                           (13, 10, 4, 90, 70),       # G = 4, N % G != 0
                                           (23, 18, 8, 70, 66),       # G = 4
                                           (70, 66, 32, 40, 36),      # G = 8, 512-thread form
                                           (700, 6, 2, 40, 30)])      # 16-column tiles (the [N][64] image exceeds the LDS)
def test_consensus_params_circulant(bk, N, d, H, P, P_hid):
    KC.check_consensus_params_circulant(bk, N, d, H, P, P_hid, S=2 if N < 100 else 1)


@pytest.mark.parametrize("S,N,B,in_dim", [(2, 3, 150, 6), (1, 7, 130, 21), (1, 2, 260, 136), (2, 9, 150, 64), (1, 16, 70, 32)])
def test_layer1_forward(bk, S, N, B, in_dim):
    KC.check_layer1_forward(bk, S, N, B, in_dim)


@pytest.mark.parametrize("S,N,B,in_dim,masked", [(2, 3, 300, 6, None), (1, 7, 130, 21, 2), (1, 2, 70, 140, None), (2, 9, 100, 32, 4), (1, 7, 45, 192, None)])
def test_sgd_fit(bk, S, N, B, in_dim, masked):
    KC.check_sgd_fit(bk, S, N, B, in_dim, steps=2, masked_agent=masked)


@pytest.mark.parametrize("S,N,B,in_dim,d,H,graph", [(2, 5, 300, 10, 4, 1, "circ"), (1, 6, 100, 18, 3, 0, "rand"),
                                                    (1, 12, 70, 24, 11, 2, "rand")])
def test_consensus_head(bk, S, N, B, in_dim, d, H, graph):
    KC.check_consensus_head(bk, S, N, B, in_dim, d, H, graph)


@pytest.mark.parametrize("S,N,B,in_dim", [(2, 3, 300, 6), (1, 5, 100, 10), (1, 8, 70, 32)])
def test_actor_step(bk, S, N, B, in_dim):
    KC.check_actor_step(bk, S, N, B, in_dim)


def test_consensus_on_shipped_reference_weights(bk, golden):
    KC.check_consensus_on_shipped_weights(bk, golden)


def test_reward_helpers(bk):
    KC.check_reward_helpers(bk, 2, 5, 300)


@pytest.mark.parametrize("S,N,nrow,ncol,mode", [(2, 5, 5, 5, "device"), (1, 70, 16, 16, "device"), (2, 5, 5, 5, "host")])
def test_rollout(bk, S, N, nrow, ncol, mode):
    KC.check_rollout(bk, S, N, nrow, ncol, steps=5, mode=mode)


@pytest.mark.parametrize("S,N,B,in_dim,advs,bs,shuffle", [(2, 5, 100, 10, [4], 32, True), (1, 5, 70, 15, [1, 3], 32, True),
                                                          (1, 2, 40, 140, [0], 32, False), (1, 6, 90, 18, [2, 5], 40, True),
                                                          (3, 5, 50, 20, [0], 7, True)])      # multi-tile batches, ragged tails
def test_minibatch_fit(bk, S, N, B, in_dim, advs, bs, shuffle):
    KC.check_minibatch_fit(bk, S, N, B, in_dim, advs, bs=bs, epochs=2, shuffle=shuffle)


@pytest.mark.parametrize("S,N,B,in_dim,advs,bs,t0", [(2, 5, 100, 10, [4], 40, 0), (1, 3, 60, 6, [0, 2], 200, 7)])
def test_minibatch_actor(bk, S, N, B, in_dim, advs, bs, t0):
    KC.check_minibatch_actor(bk, S, N, B, in_dim, advs, bs=bs, t0=t0, shuffle=B > bs)


def test_projection(bk):
    KC.check_projection(bk, 2, 3, 300, 6)


# ---- lattice (exact bf16x3) layer-1 path -----------------------------------------------------
@pytest.mark.parametrize("S,n_agents,B,width,nrow,ncol,scaling", [(2, 5, 70, 2, 5, 5, True), (1, 50, 300, 3, 32, 16, True),
                                                                    (1, 7, 33, 3, 9, 9, False)])
def test_lattice_encode(bk, S, n_agents, B, width, nrow, ncol, scaling):
    KC.check_lattice_encode(bk, S, n_agents, B, width, nrow, ncol, scaling)


@pytest.mark.parametrize("S,N,B,width,nrow,ncol", [(2, 5, 70, 2, 5, 5), (1, 13, 260, 3, 32, 32), (1, 7, 40, 2, 128, 3)])
def test_lattice_forward(bk, S, N, B, width, nrow, ncol):
    KC.check_lattice_forward(bk, S, N, B, width, nrow, ncol)


@pytest.mark.parametrize("S,N,B,width,nrow,ncol,masked", [(2, 3, 300, 2, 5, 5, None), (1, 7, 130, 3, 16, 16, 2)])
def test_lattice_sgd_fit(bk, S, N, B, width, nrow, ncol, masked):
    KC.check_lattice_sgd_fit(bk, S, N, B, width, nrow, ncol, steps=2, masked_agent=masked)


@pytest.mark.parametrize("S,N,B,in_dim,masked", [(2, 3, 300, 10, None), (1, 4, 130, 15, 2), (1, 2, 70, 32, None), (1, 3, 45, 21, None),
                                                 (1, 2, 700, 10, None)])     # 3 chunks: the multi-chunk walk of one workgroup
def test_fit_step_small(bk, S, N, B, in_dim, masked):
    KC.check_fit_step_small(bk, S, N, B, in_dim, steps=2, masked_agent=masked)


# ---- wide networks (hid != 20): dense-GEMM path, csrc/wide_kernels.hip -------------------------------
import wide_checks as WC


@pytest.mark.parametrize("S,N,B,in_dim,hid", [(2, 3, 72, 12, 32),      # float4 staging
                                              (1, 2, 150, 10, 40),     # scalar staging (unaligned K / rows), 2 n-tiles
                                              (1, 2, 40, 136, 132)])   # > 1 m-tile, k tail
def test_wide_forward(bk, S, N, B, in_dim, hid):
    WC.check_wide_forward(bk, S, N, B, in_dim, hid)


@pytest.mark.parametrize("S,N,B,in_dim,hid,masked", [(2, 3, 72, 12, 32, None), (1, 3, 70, 10, 24, 1), (1, 2, 260, 16, 64, None)])
def test_wide_fit(bk, S, N, B, in_dim, hid, masked):
    WC.check_wide_fit(bk, S, N, B, in_dim, hid, steps=2, masked_agent=masked)


@pytest.mark.parametrize("S,N,B,in_dim,hid,d,H,graph", [(2, 5, 72, 10, 32, 4, 1, "circ"), (1, 8, 140, 16, 24, 7, 2, "rand"),
                                                        (1, 24, 40, 8, 24, 23, 5, "rand")])     # no generated network: rank counting
def test_wide_consensus_head(bk, S, N, B, in_dim, hid, d, H, graph):
    WC.check_wide_consensus_head(bk, S, N, B, in_dim, hid, d, H, graph)


@pytest.mark.parametrize("S,N,B,width,nrow,ncol,masked", [(1, 5, 70, 2, 5, 5, None), (2, 11, 300, 3, 8, 6, 3)])
def test_fit_fused_lattice(bk, S, N, B, width, nrow, ncol, masked):
    KC.check_fit_fused_lattice(bk, S, N, B, width, nrow, ncol, steps=2, masked_agent=masked)


@pytest.mark.parametrize("lattice", [False, True])
def test_mid_fit_v5_matrix_core_form(bk, lattice, monkeypatch):
    """RCMARL_MIDFIT=5: k_mid_fit_v5 (layer 2 forward/backward and every row reduction on the f32 matrix core) behind the
    same two entry points, against the same oracle fits."""
    monkeypatch.setenv("RCMARL_MIDFIT", "5")
    if lattice:
        KC.check_lattice_sgd_fit(bk, 1, 5, 300, 2, 5, 5, steps=2, masked_agent=2)
    else:
        KC.check_sgd_fit(bk, 2, 5, 130, 10, steps=2, masked_agent=1)


def test_mid_fit_v3_still_reachable(bk, monkeypatch):
    """RCMARL_MIDFIT=2 keeps the VALU form (k_mid_fit_v3) behind rcmarl_mid_fit, which defaults to v5 now."""
    monkeypatch.setenv("RCMARL_MIDFIT", "2")
    KC.check_sgd_fit(bk, 1, 5, 130, 10, steps=2, masked_agent=1)


@pytest.mark.parametrize("w8", ["0", "1"])
def test_lattice_gemms_both_wavefront_shapes(bk, w8, monkeypatch):
    monkeypatch.setenv("RCMARL_LAT_W8", w8)
    KC.check_lattice_sgd_fit(bk, 1, 7, 130, 2, 5, 5, steps=2, masked_agent=2)


@pytest.mark.parametrize("S,N,B,width", [(1, 14, 300, 2)])
def test_lattice_backward_dz_fragments_from_global_bit_identical(bk, S, N, B, width, monkeypatch):
    """RCMARL_LAT_BDIRECT=1: the backward GEMM loads its three-piece operand's fragments global -> registers instead of
    staging them through LDS (lat_mainloop_bdirect).  Same products in the same order: the same oracle fit, and weights
    and next-step operand pieces equal to the LDS-staged kernel's bit for bit."""
    ref_msg, ref_wp = KC.check_lattice_sgd_fit(bk, S, N, B, width, 7, 9, steps=2, masked_agent=4)
    monkeypatch.setenv("RCMARL_LAT_BDIRECT", "1")
    msg, wp = KC.check_lattice_sgd_fit(bk, S, N, B, width, 7, 9, steps=2, masked_agent=4)
    np.testing.assert_array_equal(msg, ref_msg)
    np.testing.assert_array_equal(wp, ref_wp)


@pytest.mark.parametrize("d,H", [(4, 1), (6, 2), (10, 4), (18, 8), (5, 1), (9, 3)])
def test_consensus_params_bits_on_awkward_data(bk, d, H):
    """K1 (both kernels on circulant graphs) against a plain NumPy statement of its arithmetic, BIT FOR BIT, on columns of
    zeros, subnormal-range sums (the guarded path of the constant division), 1e30-scale values, exact ties, a random
    cooperation mask and agent counts that are not a multiple of the kernel's agent group; seeds and sizes drawn here."""
    rng = np.random.default_rng(1000 * d + H)
    for _ in range(3):
        N = int(rng.integers(d, 40))
        P_hid = int(rng.integers(1, 150))
        graph = "circ" if rng.random() < 0.7 else "rand"
        KC.check_consensus_params_exact(bk, N, d, H, P_hid, int(rng.integers(1, 3)), int(rng.integers(1 << 30)), graph)


def test_lattice_gemms_spread_dma_issue_bit_identical(bk, monkeypatch):
    """RCMARL_LAT_SPREAD=3: both lattice GEMMs issue the LDS-DMA bursts of the next k-tile between their matrix-core
    instructions instead of back to back after the barrier.  Pure scheduling: same bits."""
    args = (1, 14, 300, 2, 7, 9)
    monkeypatch.setenv("RCMARL_LAT_SPREAD", "0")
    ref_msg, ref_wp = KC.check_lattice_sgd_fit(bk, *args, steps=2, masked_agent=4)
    monkeypatch.setenv("RCMARL_LAT_SPREAD", "3")
    msg, wp = KC.check_lattice_sgd_fit(bk, *args, steps=2, masked_agent=4)
    np.testing.assert_array_equal(msg, ref_msg)
    np.testing.assert_array_equal(wp, ref_wp)


@pytest.mark.parametrize("knob", ["RCMARL_LAT_WIDE", "RCMARL_LAT_TALL"])
def test_lattice_backward_wide_tile_bit_identical(bk, knob, monkeypatch):
    """The backward GEMM on 256 x 256 tiles (RCMARL_LAT_WIDE: the one-piece operand's LDS stage shared by twice the dz
    columns) or 512 x 128 tiles (RCMARL_LAT_TALL: the dz panel read once for up to 512 inputs), eight wavefronts each.
    Same products in the same order per accumulator: same bits."""
    args = (1, 14, 300, 2, 7, 9)
    ref_msg, ref_wp = KC.check_lattice_sgd_fit(bk, *args, steps=2, masked_agent=4)
    monkeypatch.setenv(knob, "1")
    msg, wp = KC.check_lattice_sgd_fit(bk, *args, steps=2, masked_agent=4)
    np.testing.assert_array_equal(msg, ref_msg)
    np.testing.assert_array_equal(wp, ref_wp)


def test_mid_fit_4x4_block_products_bit_identical(bk, monkeypatch):
    """RCMARL_MIDFIT=6: the two 20x20 layer products of the mid kernel as 4x4x1 sixteen-block MFMAs (result born row-per-lane,
    no padding rows, no permlane swaps).  Same fmaf chains: dz pieces AND gradient records equal v5's bit for bit, on both
    entry points; and the usual oracle fit."""
    from rcmarl_amd import lattice as LT
    rng = np.random.default_rng(4)
    S, N, B, in_dim = (1, 3, 300, 6)
    P, _ = KC.geom(in_dim, 1)
    ldp, ldb = KC.pad64(P), KC.pad64(B)
    theta = KC.pack_rows(KC.random_params(rng, S, N, in_dim, 1), ldp)
    a1 = np.maximum(rng.normal(size=(S, N * 20, ldb)), 0.1 * rng.normal(size=(S, N * 20, ldb))).astype(np.float32)
    y = rng.normal(size=(S, N, ldb)).astype(np.float32)
    g = LT.Geometry(N, in_dim, B)
    nchunk, psz = (B + 255) // 256, bk.lib.rcmarl_fit_partial_size(20)
    out = {}
    for var in ("5", "6"):
        monkeypatch.setenv("RCMARL_MIDFIT", var)
        d_a, d_th, d_y = bk.dev(a1), bk.dev(theta), bk.dev(y)
        d_part = bk.dev(np.zeros((S, N, nchunk, psz), np.float32))
        d_dzp = bk.dev(np.zeros(S * LT.Geometry.nbytes(g.dzp, 3) // 2, np.uint16))
        bk.lib.rcmarl_mid_fit_lattice(bk.ptr(d_a), bk.ptr(d_th), bk.ptr(d_y), bk.ptr(d_part), bk.ptr(d_dzp), g.dzp[0], g.dzp[1], S, N, B,
                                      in_dim, 20, ldp, ldb, bk.stream)
        d_a2 = bk.dev(a1)
        d_part2 = bk.dev(np.zeros((S, N, nchunk, psz), np.float32))
        bk.lib.rcmarl_mid_fit(bk.ptr(d_a2), bk.ptr(d_th), bk.ptr(d_y), bk.ptr(d_part2), S, N, B, in_dim, 20, ldp, ldb, bk.stream)
        out[var] = (bk.host(d_dzp).copy(), bk.host(d_part).copy(), bk.host(d_a2).copy(), bk.host(d_part2).copy())
    for a, b in zip(out["5"], out["6"]):
        np.testing.assert_array_equal(a, b)
    KC.check_sgd_fit(bk, 1, 5, 130, 10, steps=2, masked_agent=1)

