"""Parity tests proper: every HIP kernel, on a real MI355X, through the product
C-ABI (librcmarl_hip.so), against the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest
import torch

import kernel_checks as KC

pytestmark = pytest.mark.gpu


class GpuBackend:
    def __init__(self):
        from rcmarl_amd import capi
        self.lib = capi.load()
        assert torch.cuda.is_available(), "GPU tests need a ROCm device"

    @property
    def stream(self):
        return torch.cuda.current_stream().cuda_stream

    def dev(self, arr):
        arr = np.ascontiguousarray(arr)
        if arr.dtype == np.uint64:
            return torch.from_numpy(arr.view(np.int64)).cuda()
        return torch.from_numpy(arr).cuda()

    def ptr(self, h):
        return None if h is None else h.data_ptr()

    def host(self, h):
        torch.cuda.synchronize()
        return h.cpu().numpy()


@pytest.fixture(scope="module")
def bk():
    return GpuBackend()


@pytest.mark.parametrize("N,d,H,P,P_hid,graph", [
    (5, 4, 1, 661, 640, "circ"), (5, 4, 0, 761, 740, "circ"), (64, 10, 4, 3021, 3000, "rand"),
    (256, 18, 8, 1100, 1070, "circ"), (256, 4, 1, 700, 650, "circ"), (300, 18, 8, 300, 280, "rand"),
    (30, 23, 5, 90, 77, "rand"), (100, 66, 32, 200, 190, "rand"), (1024, 18, 8, 200, 130, "circ"),
])
def test_consensus_params(bk, N, d, H, P, P_hid, graph):
    KC.check_consensus_params(bk, N, d, H, P, P_hid, graph)


@pytest.mark.parametrize("N,d,H,P,P_hid,S", [(5, 4, 1, 761, 740, 300), (64, 10, 4, 3021, 3000, 2), (256, 18, 8, 1100, 1070, 2),
                                             (256, 18, 8, 10701, 10680, 3), (37, 34, 16, 300, 280, 2), (100, 66, 32, 200, 190, 2),
                                             (1024, 66, 32, 300, 260, 1), (1024, 18, 8, 200, 130, 2), (300, 6, 2, 700, 650, 3)])
def test_consensus_params_circulant(bk, N, d, H, P, P_hid, S):
    KC.check_consensus_params_circulant(bk, N, d, H, P, P_hid, S=S)


@pytest.mark.parametrize("S,N,B,in_dim", [(2, 5, 1000, 10), (1, 64, 700, 128), (2, 7, 130, 21), (1, 13, 3000, 39), (2, 64, 3000, 192), (1, 256, 1000, 512), (1, 9, 150, 64)])
def test_layer1_forward(bk, S, N, B, in_dim):
    KC.check_layer1_forward(bk, S, N, B, in_dim)


@pytest.mark.parametrize("S,N,B,in_dim,masked", [(2, 5, 1000, 10, None), (2, 5, 3000, 15, 4), (1, 32, 700, 64, 2),
                                                 (1, 7, 130, 21, None), (1, 64, 1000, 192, 5), (1, 128, 333, 256, None)])
def test_sgd_fit(bk, S, N, B, in_dim, masked):
    KC.check_sgd_fit(bk, S, N, B, in_dim, steps=5, masked_agent=masked)


@pytest.mark.parametrize("S,N,B,in_dim,d,H,graph", [(2, 5, 1000, 10, 4, 1, "circ"), (1, 24, 700, 48, 10, 4, "rand"),
                                                    (1, 20, 300, 40, 18, 8, "circ"), (1, 12, 70, 24, 11, 2, "rand"),
                                                    (1, 30, 300, 90, 23, 5, "rand")])
def test_consensus_head(bk, S, N, B, in_dim, d, H, graph):
    KC.check_consensus_head(bk, S, N, B, in_dim, d, H, graph)


@pytest.mark.parametrize("S,N,B,in_dim", [(2, 5, 1000, 10), (1, 16, 300, 32), (1, 64, 1000, 128)])
def test_actor_step(bk, S, N, B, in_dim):
    KC.check_actor_step(bk, S, N, B, in_dim, steps=3)


def test_consensus_on_shipped_reference_weights(bk, golden):
    KC.check_consensus_on_shipped_weights(bk, golden)


def test_reward_helpers(bk):
    KC.check_reward_helpers(bk, 2, 5, 1000)
    KC.check_reward_helpers(bk, 1, 64, 3000)


@pytest.mark.parametrize("S,N,nrow,ncol,mode", [(3, 5, 5, 5, "device"), (2, 70, 16, 16, "device"), (2, 5, 5, 5, "host")])
def test_rollout(bk, S, N, nrow, ncol, mode):
    KC.check_rollout(bk, S, N, nrow, ncol, steps=20, mode=mode)


@pytest.mark.parametrize("S,N,B,in_dim,advs,bs,shuffle", [(2, 5, 1000, 10, [4], 32, True), (1, 5, 3000, 15, [1, 3], 32, True),
                                                          (1, 64, 500, 192, [0, 63], 32, True), (1, 256, 200, 768, [7], 32, False),
                                                          (7, 6, 900, 18, [2, 5], 40, True), (9, 5, 333, 20, [0], 7, True),
                                                          (300, 5, 400, 15, [4], 32, True)])
def test_minibatch_fit(bk, S, N, B, in_dim, advs, bs, shuffle):
    KC.check_minibatch_fit(bk, S, N, B, in_dim, advs, bs=bs, epochs=3, shuffle=shuffle)


@pytest.mark.parametrize("S,N,B,in_dim,advs,bs,t0", [(2, 5, 1000, 10, [4], 200, 0), (1, 5, 1000, 10, [0, 2], 200, 15),
                                                     (1, 64, 400, 128, [5], 200, 3)])
def test_minibatch_actor(bk, S, N, B, in_dim, advs, bs, t0):
    KC.check_minibatch_actor(bk, S, N, B, in_dim, advs, bs=bs, t0=t0, shuffle=True)


def test_projection(bk):
    KC.check_projection(bk, 2, 5, 1000, 10)
    KC.check_projection(bk, 1, 16, 700, 48)


# ---- lattice (exact bf16x3) layer-1 path -----------------------------------------------------
@pytest.mark.parametrize("S,n_agents,B,width,nrow,ncol,scaling", [(2, 5, 1000, 2, 5, 5, True), (1, 64, 700, 3, 16, 16, True),
                                                                    (2, 256, 3000, 3, 32, 32, True), (1, 7, 33, 3, 9, 9, False)])
def test_lattice_encode(bk, S, n_agents, B, width, nrow, ncol, scaling):
    KC.check_lattice_encode(bk, S, n_agents, B, width, nrow, ncol, scaling)


@pytest.mark.parametrize("S,N,B,width,nrow,ncol", [(2, 5, 1000, 2, 5, 5), (1, 64, 700, 3, 16, 16), (8, 64, 3000, 2, 16, 16),
                                                   (1, 256, 1000, 2, 32, 32), (1, 13, 3000, 3, 128, 7)])
def test_lattice_forward(bk, S, N, B, width, nrow, ncol):
    KC.check_lattice_forward(bk, S, N, B, width, nrow, ncol)


@pytest.mark.parametrize("S,N,B,width,nrow,ncol,masked", [(2, 5, 1000, 2, 5, 5, None), (2, 5, 3000, 3, 5, 5, 4),
                                                          (1, 64, 1000, 3, 16, 16, 5), (8, 32, 700, 2, 16, 16, 2),
                                                          (1, 128, 333, 2, 32, 32, None)])
def test_lattice_sgd_fit(bk, S, N, B, width, nrow, ncol, masked):
    KC.check_lattice_sgd_fit(bk, S, N, B, width, nrow, ncol, steps=5, masked_agent=masked)


@pytest.mark.parametrize("width", [2, 3])
def test_lattice_equals_f32_path_at_full_size(bk, width):
    """BASELINE configs[3] sizes (N=256 agents, B=3000 rows, 32x32 grid): the bf16x3 lattice GEMMs and the
    f32-MFMA GEMMs are independent implementations of the same fp32 math -> they must agree to fp32 roundoff
    (forward activations, and W1 after two full SGD steps through mid_fit)."""
    KC.check_lattice_vs_f32(bk, S=2, N=256, B=3000, width=width, nrow=32, ncol=32, steps=2)


@pytest.mark.parametrize("S,N,B,in_dim,masked", [(2, 5, 1000, 10, None), (2, 5, 3000, 15, 4), (4, 5, 700, 15, None), (1, 10, 333, 30, 2),
                                                 (3, 16, 1000, 32, None), (1, 8, 130, 24, None)])
def test_fit_step_small(bk, S, N, B, in_dim, masked):
    KC.check_fit_step_small(bk, S, N, B, in_dim, steps=5, masked_agent=masked)


@pytest.mark.parametrize("N,d,H,P,P_hid,S", [(5, 4, 1, 761, 740, 512), (5, 4, 0, 661, 640, 300), (12, 6, 2, 1200, 1100, 128)])
def test_consensus_params_short_tiles_many_seeds(bk, N, d, H, P, P_hid, S):
    """Regression: with few agents and a small d a tile is aggregated in less time than the LDS-DMA of the next one
    needs; the persistent walk must wait for it (K1 v2 once read stale LDS here: wrong, run-to-run different results).
    Checked against the oracle and for bit-identical repetition."""
    KC.check_consensus_params(bk, N, d, H, P, P_hid, "circ", S=S)
    KC.check_consensus_params(bk, N, d, H, P, P_hid, "circ", S=S)


# ---- wide networks (hid != 20): dense-GEMM path, csrc/wide_kernels.hip -------------------------------
import wide_checks as WC


@pytest.mark.parametrize("S,N,B,in_dim,hid", [(2, 5, 1000, 12, 64), (1, 3, 3000, 10, 40), (1, 4, 700, 256, 512), (1, 2, 333, 136, 132)])
def test_wide_forward(bk, S, N, B, in_dim, hid):
    WC.check_wide_forward(bk, S, N, B, in_dim, hid)


@pytest.mark.parametrize("S,N,B,in_dim,hid,masked", [(2, 5, 1000, 12, 64, None), (1, 3, 701, 10, 24, 1), (1, 4, 3000, 128, 512, 2),
                                                     (1, 16, 1000, 32, 128, None)])
def test_wide_fit(bk, S, N, B, in_dim, hid, masked):
    WC.check_wide_fit(bk, S, N, B, in_dim, hid, steps=3, masked_agent=masked)


@pytest.mark.parametrize("S,N,B,in_dim,hid,d,H,graph", [(2, 5, 1000, 10, 64, 4, 1, "circ"), (1, 24, 700, 48, 128, 10, 4, "rand"),
                                                        (1, 70, 300, 140, 512, 66, 32, "circ"), (1, 30, 300, 60, 64, 23, 5, "rand")])
def test_wide_consensus_head(bk, S, N, B, in_dim, hid, d, H, graph):
    WC.check_wide_consensus_head(bk, S, N, B, in_dim, hid, d, H, graph)


@pytest.mark.parametrize("S,N,B,width,nrow,ncol,masked", [(2, 5, 1000, 2, 5, 5, None), (2, 11, 300, 3, 8, 6, 3), (1, 64, 3000, 3, 16, 16, 5),
                                                          (2, 256, 1000, 2, 32, 32, None), (1, 256, 3000, 3, 32, 32, 100),
                                                          (1, 20, 777, 2, 7, 9, None)])
def test_fit_fused_lattice(bk, S, N, B, width, nrow, ncol, masked):
    """The fused local-fit step (csrc/lattice_fit.hip) vs the unfused pair (dz1 bit-identical) and vs the oracle's fit."""
    # N = 256: fast_lr 0.0025 as everywhere at that size (0.01 is on the edge of divergence at 512/768 unscaled inputs and
    # amplifies fp32 roundoff between any two summation orders; bench.py header)
    KC.check_fit_fused_lattice(bk, S, N, B, width, nrow, ncol, steps=3, masked_agent=masked, lr=0.0025 if N >= 256 else 0.01)


@pytest.mark.parametrize("S,N,B,in_dim,masked", [(2, 5, 1000, 10, None), (2, 5, 3000, 15, 4), (1, 64, 1000, 192, 5), (1, 128, 333, 256, None)])
def test_mid_fit_v5_sgd_fit(bk, S, N, B, in_dim, masked, monkeypatch):
    """RCMARL_MIDFIT=5: the all-matrix-core form of the mid kernel behind rcmarl_mid_fit, same oracle fits as test_sgd_fit."""
    monkeypatch.setenv("RCMARL_MIDFIT", "5")
    KC.check_sgd_fit(bk, S, N, B, in_dim, steps=5, masked_agent=masked)


@pytest.mark.parametrize("S,N,B,width,nrow,ncol,masked", [(2, 5, 1000, 2, 5, 5, None), (1, 64, 1000, 3, 16, 16, 5), (8, 32, 700, 2, 16, 16, 2), (1, 20, 777, 2, 7, 9, 3)])
def test_mid_fit_v5_lattice_sgd_fit(bk, S, N, B, width, nrow, ncol, masked, monkeypatch):
    monkeypatch.setenv("RCMARL_MIDFIT", "5")
    KC.check_lattice_sgd_fit(bk, S, N, B, width, nrow, ncol, steps=5, masked_agent=masked)


def test_mid_fit_v5_bit_identical_dz_to_v3(bk, monkeypatch):
    """Same fmaf chains in both forms: the dz1 pieces of v5 equal those of v3 bit for bit (records: summation order)."""
    import torch
    rng = np.random.default_rng(3)
    S, N, B, in_dim = 2, 6, 700, 12
    P, _ = KC.geom(in_dim, 1)
    ldp, ldb = KC.pad64(P), KC.pad64(B)
    theta = KC.pack_rows(KC.random_params(rng, S, N, in_dim, 1), ldp)
    a1 = np.maximum(rng.normal(size=(S, N * 20, ldb)), 0.1 * rng.normal(size=(S, N * 20, ldb))).astype(np.float32)
    y = rng.normal(size=(S, N, ldb)).astype(np.float32)
    from rcmarl_amd import lattice as LT
    g = LT.Geometry(N, in_dim, B)
    nchunk, psz = (B + 255) // 256, bk.lib.rcmarl_fit_partial_size(20)
    out = {}
    for var in ("2", "5"):
        monkeypatch.setenv("RCMARL_MIDFIT", var)
        d_a, d_th, d_y = bk.dev(a1), bk.dev(theta), bk.dev(y)
        d_part = bk.dev(np.zeros((S, N, nchunk, psz), np.float32))
        d_dzp = bk.dev(np.zeros(S * LT.Geometry.nbytes(g.dzp, 3) // 2, np.uint16))
        bk.lib.rcmarl_mid_fit_lattice(bk.ptr(d_a), bk.ptr(d_th), bk.ptr(d_y), bk.ptr(d_part), bk.ptr(d_dzp), g.dzp[0], g.dzp[1], S, N, B,
                                      in_dim, 20, ldp, ldb, bk.stream)
        out[var] = (bk.host(d_dzp).copy(), bk.host(d_part).copy())
    np.testing.assert_array_equal(out["5"][0], out["2"][0])
    pu, pf = out["2"][1], out["5"][1]
    scale = np.maximum(np.abs(pu).max(axis=(2, 3), keepdims=True), 1e-6)
    assert float((np.abs(pf - pu) / scale).max()) <= 2e-5


@pytest.mark.parametrize("variant", ["2", "1", "0"])
def test_mid_fit_older_variants(bk, variant, monkeypatch):
    """The earlier forms of the mid kernel stay selectable (RCMARL_MIDFIT) and correct."""
    monkeypatch.setenv("RCMARL_MIDFIT", variant)
    KC.check_sgd_fit(bk, 2, 5, 1000, 10, steps=3, masked_agent=None)


@pytest.mark.parametrize("w8", ["0", "1"])
def test_lattice_gemms_both_wavefront_shapes(bk, w8, monkeypatch):
    """RCMARL_LAT_W8: the lattice GEMMs as four wavefronts of 64x128 / 128x64 or eight of 64x64 per workgroup -- same
    products, same k order: the same oracle fit either way (forward defaults to eight, backward to four)."""
    monkeypatch.setenv("RCMARL_LAT_W8", w8)
    KC.check_lattice_sgd_fit(bk, 2, 20, 777, 3, 7, 9, steps=3, masked_agent=4)
    KC.check_lattice_forward(bk, 1, 64, 1000, 2, 16, 16)


@pytest.mark.parametrize("S,N,B,width", [(2, 20, 777, 3), (1, 40, 300, 2)])
def test_lattice_backward_dz_fragments_from_global_bit_identical(bk, S, N, B, width, monkeypatch):
    """RCMARL_LAT_BDIRECT=1: the backward GEMM loads its three-piece operand's fragments global -> registers instead of
    staging them through LDS (lat_mainloop_bdirect).  Same products in the same order: the same oracle fit, and weights
    and next-step operand pieces equal to the LDS-staged kernel's bit for bit."""
    ref_msg, ref_wp = KC.check_lattice_sgd_fit(bk, S, N, B, width, 7, 9, steps=3, masked_agent=4)
    monkeypatch.setenv("RCMARL_LAT_BDIRECT", "1")
    msg, wp = KC.check_lattice_sgd_fit(bk, S, N, B, width, 7, 9, steps=3, masked_agent=4)
    np.testing.assert_array_equal(msg, ref_msg)
    np.testing.assert_array_equal(wp, ref_wp)


@pytest.mark.parametrize("d,H", [(4, 1), (6, 2), (10, 4), (18, 8), (5, 1), (9, 3)])
def test_consensus_params_bits_on_awkward_data(bk, d, H):
    """K1 (both kernels on circulant graphs) against a plain NumPy statement of its arithmetic, BIT FOR BIT, on columns of
    zeros, subnormal-range sums (the guarded path of the constant division), 1e30-scale values, exact ties, a random
    cooperation mask and agent counts that are not a multiple of the kernel's agent group; seeds and sizes drawn here."""
    rng = np.random.default_rng(1000 * d + H)
    for _ in range(6):
        N = int(rng.integers(d, 300))
        P_hid = int(rng.integers(1, 700))
        graph = "circ" if rng.random() < 0.7 else "rand"
        KC.check_consensus_params_exact(bk, N, d, H, P_hid, int(rng.integers(1, 3)), int(rng.integers(1 << 30)), graph)


def test_lattice_gemms_spread_dma_issue_bit_identical(bk, monkeypatch):
    """RCMARL_LAT_SPREAD=3: both lattice GEMMs issue the LDS-DMA bursts of the next k-tile between their matrix-core
    instructions instead of back to back after the barrier.  Pure scheduling: same bits."""
    args = (2, 20, 777, 3, 7, 9)
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
    args = (2, 26, 777, 3, 7, 9)
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
    S, N, B, in_dim = (2, 6, 700, 12)
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
    KC.check_sgd_fit(bk, 2, 5, 1000, 10, steps=2, masked_agent=1)

