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
@pytest.mark.parametrize("mx", ["1", "0"])       # layer 2 + heads on the f16 matrix core (k_consensus_head_mx) | everything on the vector ALUs
def test_consensus_head(bk, S, N, B, in_dim, d, H, graph, mx, monkeypatch):
    monkeypatch.setenv("RCMARL_K2_MX", mx)
    KC.check_consensus_head(bk, S, N, B, in_dim, d, H, graph)


@pytest.mark.parametrize("cpw", ["2", "3"])
def test_consensus_head_matrix_core_form_several_chunks_per_workgroup(bk, cpw, monkeypatch):
    """k_consensus_head_mx walks several 256-row chunks per workgroup when there are many (seed, agent) pairs: forced here
    (RCMARL_K2_CPW) on three chunks, the last one ragged -- one record per workgroup, zeros in the other chunks' slots."""
    monkeypatch.setenv("RCMARL_K2_MX", "1")
    monkeypatch.setenv("RCMARL_K2_CPW", cpw)
    KC.check_consensus_head(bk, 1, 5, 600, 10, 4, 1, "circ")
    monkeypatch.setenv("RCMARL_K2_CPW", "2")
    KC.check_consensus_head(bk, 1, 5, 600, 10, 4, 1, "circ", outlier=1e4, compare=False)     # (the fp32 lane code behind the loop)


def test_consensus_head_out_of_range_head_takes_the_fp32_lane_code(bk, monkeypatch):
    """A message head beyond the f16 range of the matrix-core form (2^10 |W3| > 65000): the workgroups that see it run the fp32 lane
    code inside k_consensus_head_mx: the bits of k_consensus_head (which test_consensus_head holds to the oracle)."""
    res = {}
    for mx in ("1", "0"):
        monkeypatch.setenv("RCMARL_K2_MX", mx)
        res[mx] = KC.check_consensus_head(bk, 2, 5, 300, 10, 4, 1, "circ", outlier=1e4, compare=False)
    sees = [1, 3, 4]                             # the agents whose in-neighbourhood [i, i+1, i+2, i+3] holds agent 1, the outlier
    np.testing.assert_array_equal(res["1"][0][:, sees], res["0"][0][:, sees])
    np.testing.assert_array_equal(res["1"][1][:, sees], res["0"][1][:, sees])
    assert not np.array_equal(res["1"][0][:, 2], res["0"][0][:, 2])        # (agent 2 does not: matrix-core form, other last bits)


def test_consensus_head_out_of_range_activations_take_the_fp32_lane_code(bk, monkeypatch):
    """layer-1 activations beyond the f16 range in one replay row: the 64 rows of that wavefront (and only those) carry the aggregate of
    the fp32 lane code, bit for bit (the slow path of k_consensus_head_mx behind its chunk loop, decided per wavefront by a ballot)"""
    res = {}
    for mx in ("1", "0"):
        monkeypatch.setenv("RCMARL_K2_MX", mx)
        res[mx] = KC.check_consensus_head(bk, 1, 5, 600, 10, 4, 1, "circ", outlier=1.0, compare=False, big_x_rows=[300])
    a, b = res["1"][1], res["0"][1]                                            # the aggregates [S][N][ldb]
    np.testing.assert_array_equal(a[:, 1:, 256:320], b[:, 1:, 256:320])
    assert not np.array_equal(a[:, 1:, :256], b[:, 1:, :256])
    np.testing.assert_allclose(a[:, 1:, :256], b[:, 1:, :256], rtol=0, atol=3e-5 * max(1.0, float(np.abs(b[:, 1:, :256]).max())))
    np.testing.assert_allclose(res["1"][0], res["0"][0], rtol=0, atol=2e-5 * max(1.0, float(np.abs(res["0"][0]).max())))


@pytest.mark.parametrize("mx", ["1", "0"])       # layer 2 on the f16 matrix core (k_mid_value_mx) | on the vector ALUs (k_mid_value)
@pytest.mark.parametrize("S,N,B,in_dim,row_off", [(2, 3, 300, 6, 0), (1, 5, 700, 10, 1), (1, 2, 64, 4, 0)])
def test_mid_value(bk, S, N, B, in_dim, row_off, mx, monkeypatch):
    monkeypatch.setenv("RCMARL_MIDVALUE_MX", mx)
    KC.check_mid_value(bk, S, N, B, in_dim, row_off=row_off)


def test_mid_value_f32_entry_is_the_vector_alu_kernel(bk, monkeypatch):
    """rcmarl_mid_value_f32 (the adversaries' callers): the bits of rcmarl_mid_value with RCMARL_MIDVALUE_MX=0, whatever the knob says"""
    monkeypatch.setenv("RCMARL_MIDVALUE_MX", "0")
    want = KC.check_mid_value(bk, 1, 3, 300, 6, compare=False)
    monkeypatch.setenv("RCMARL_MIDVALUE_MX", "1")
    got = KC.check_mid_value(bk, 1, 3, 300, 6, compare=False, entry="rcmarl_mid_value_f32")
    for with_r in (True, False):
        np.testing.assert_array_equal(got[with_r], want[with_r])


def test_mid_value_out_of_range_rows_take_the_fp32_lane_code(bk, monkeypatch):
    """weights beyond the f16 range of the matrix-core form (a whole agent) or activations beyond it (the 64 rows of one wavefront):
    the fp32 lane code inside k_mid_value_mx -- the bits of k_mid_value; everything else keeps the matrix-core form's last bits"""
    res = {}
    for mx in ("1", "0"):
        monkeypatch.setenv("RCMARL_MIDVALUE_MX", mx)
        res[mx] = KC.check_mid_value(bk, 1, 3, 600, 6, big_w2_agent=1, big_a1_rows=[300], compare=False)
    for with_r in (True, False):
        a, b = res["1"][with_r], res["0"][with_r]
        np.testing.assert_array_equal(a[:, 1], b[:, 1])                        # agent 1: W2 out of range
        np.testing.assert_array_equal(a[:, 0, 256:320], b[:, 0, 256:320])      # agent 0, the wavefront that holds row 300
        assert not np.array_equal(a[:, 0, :256], b[:, 0, :256])                # the rest of agent 0: matrix-core form
        np.testing.assert_allclose(a[:, 0, :256], b[:, 0, :256], rtol=0, atol=3e-6 * max(1.0, float(np.abs(b[:, 0, :256]).max())))


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


@pytest.mark.parametrize("S,N,B,in_dim,advs,bs,shuffle", [(1, 5, 70, 15, [1, 3], 32, True), (3, 5, 50, 20, [0], 7, True)])
def test_minibatch_fit_fp32_wavefront_kernel(bk, S, N, B, in_dim, advs, bs, shuffle, monkeypatch):
    """RCMARL_MB_MX=0: k_minibatch_wave alone instead of k_minibatch_mx (f16 matrix core) + fix-up"""
    monkeypatch.setenv("RCMARL_MB_MX", "0")
    KC.check_minibatch_fit(bk, S, N, B, in_dim, advs, bs=bs, epochs=2, shuffle=shuffle)


@pytest.mark.parametrize("S,N,B,in_dim,advs,bs,shuffle", [(2, 5, 100, 10, [4], 32, True), (1, 5, 70, 15, [1, 3], 32, True), (3, 5, 50, 20, [0], 7, True), (1, 6, 90, 18, [2, 5], 90, False)])
def test_minibatch_fit_compact_form(bk, S, N, B, in_dim, advs, bs, shuffle, monkeypatch):
    """RCMARL_MB_MX_COMPACT=1: k_minibatch_mx with the short k-step-1 weight fragments and ONE B plane pair for both gradient
    products (eight wavefronts per CU; the default from 1537 networks per launch on), mini-batches and full batches (bs = B:
    the cooperative agents' local fit, engine._fit_as_chains)."""
    monkeypatch.setenv("RCMARL_MB_MX_COMPACT", "1")
    KC.check_minibatch_fit(bk, S, N, B, in_dim, advs, bs=bs, epochs=2, shuffle=shuffle, knife_edge_nets=1 if B >= 900 else 0)


@pytest.mark.parametrize("seeds,calls,epochs,B", [((3, 77), (0, 5), 2, 70), ((11,), (2,), 1, 1), ((1000,), (0, 1, 2), 2, 300)])
def test_shuffle_perms_equal_the_oracle_stream(bk, seeds, calls, epochs, B):
    KC.check_shuffle_perms(bk, seeds, calls, epochs, B)


def test_minibatch_fit_out_of_range_network_is_redone_in_fp32(bk, monkeypatch):
    import numpy as np
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("RCMARL_MB_MX", mode)
        res[mode] = KC.run_minibatch_fit_with_blown_network(bk)
    np.testing.assert_array_equal(res["1"][0], res["0"][0])
    assert not np.array_equal(res["1"][1], res["0"][1])
    assert np.abs(res["1"][1] - res["0"][1]).max() <= 1e-5


@pytest.mark.parametrize("S,N,B,in_dim,advs,bs,t0", [(2, 5, 100, 10, [4], 40, 0), (1, 3, 60, 6, [0, 2], 200, 7)])
def test_minibatch_actor(bk, S, N, B, in_dim, advs, bs, t0):
    KC.check_minibatch_actor(bk, S, N, B, in_dim, advs, bs=bs, t0=t0, shuffle=B > bs)


@pytest.mark.parametrize("in_dims,blow,compact", [((10, 15, 10), None, None), ((18, 20, 18), None, None),      # both input classes
                                                  ((10, 15, 10), (1, 0), None), ((18, 20, 18), (2, 1), "1")])    # a flag set by job j > 0
def test_minibatch_fit_multi_equals_single_job_launches(bk, in_dims, blow, compact, monkeypatch):
    if compact is not None:
        monkeypatch.setenv("RCMARL_MB_MX_COMPACT", compact)
    KC.check_minibatch_fit_multi(bk, in_dims=in_dims, blow=blow)


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


def test_mid_step_f16_kernel_vs_fp32_kernel(bk):
    KC.check_mid_step_f16_vs_fp32_kernel(bk, 2, 6, 300, 12)


# ---- wide networks (hid != 20): dense-GEMM path, csrc/wide_kernels.hip -------------------------------
import wide_checks as WC


@pytest.mark.parametrize("S,N,B,in_dim,hid", [(2, 3, 72, 12, 32),      # float4 staging
                                              (1, 2, 150, 10, 40),     # scalar staging (unaligned K / rows), 2 n-tiles
                                              (1, 2, 40, 136, 132)])   # > 1 m-tile, k tail
@pytest.mark.parametrize("f16", [1, 0])
def test_wide_forward(bk, S, N, B, in_dim, hid, f16, wide_form):
    wide_form(bk, f16)
    WC.check_wide_forward(bk, S, N, B, in_dim, hid)


@pytest.mark.parametrize("S,N,B,in_dim,hid,masked", [(2, 3, 72, 12, 32, None), (1, 3, 70, 10, 24, 1), (1, 2, 260, 16, 64, None)])
@pytest.mark.parametrize("f16", [1, 0])
def test_wide_fit(bk, S, N, B, in_dim, hid, masked, f16, wide_form):
    wide_form(bk, f16)
    WC.check_wide_fit(bk, S, N, B, in_dim, hid, steps=2, masked_agent=masked)


@pytest.mark.parametrize("S,N,B,hid", [(2, 3, 70, 24), (1, 5, 200, 32)])
def test_pack_dz_with_row_sums(bk, S, N, B, hid):
    KC.check_pack_dz_rowsum(bk, S, N, B, hid)


def test_wide_dense_layer_out_of_f16_range_recomputes_in_fp32(bk):
    WC.check_wide_out_of_range(bk, *(1, 3, 72, 12, 32))


@pytest.mark.parametrize("S,N,B,in_dim,hid,d,H,graph", [(2, 5, 72, 10, 32, 4, 1, "circ"), (1, 8, 140, 16, 24, 7, 2, "rand"),
                                                        (1, 24, 40, 8, 24, 23, 5, "rand")])     # no generated network: rank counting
@pytest.mark.parametrize("f16", [1, 0])
def test_wide_consensus_head(bk, S, N, B, in_dim, hid, d, H, graph, f16, wide_form):
    wide_form(bk, f16)
    WC.check_wide_consensus_head(bk, S, N, B, in_dim, hid, d, H, graph)


# ---- the same layers on pre-split packed operands (hid % 128 == 0): csrc/dense_pk.hip ---------------------------------
@pytest.mark.parametrize("S,N,B,width,hid", [(1, 2, 150, 2, 128), (1, 1, 300, 3, 256)])      # 128 x 128 tiles / 256 x 256 tiles
def test_pk_forward(bk, S, N, B, width, hid):
    WC.check_pk_forward(bk, S, N, B, width, 5, 5, hid)


@pytest.mark.parametrize("S,N,B,width,nrow,ncol,hid,steps,masked", [(1, 2, 300, 2, 5, 5, 256, 3, None),      # two unit tiles, three row tiles
                                                                    (2, 3, 150, 3, 7, 9, 128, 2, 1)])
def test_pk_fit(bk, S, N, B, width, nrow, ncol, hid, steps, masked):
    WC.check_pk_fit(bk, S, N, B, width, nrow, ncol, hid, steps=steps, lr=0.05, masked_agent=masked, tol=1e-6)


def test_pk_operands_beyond_the_f16_range_raise_the_flag(bk):
    WC.check_pk_range_flag(bk)


@pytest.mark.parametrize("m128", ["0", "1"])
def test_lattice_backward_both_tile_heights(bk, m128, monkeypatch):
    """Networks of at most 128 inputs take 128-row tiles in the backward GEMM (RCMARL_LAT_M128=0: the 256-row tile of the wide
    inputs): the same products in the same k order, the same oracle fit either way."""
    monkeypatch.setenv("RCMARL_LAT_M128", m128)
    KC.check_lattice_sgd_fit(bk, 1, 7, 130, 2, 5, 5, steps=2, masked_agent=2)


@pytest.mark.parametrize("w8", ["0", "1"])
def test_lattice_gemms_both_wavefront_shapes(bk, w8, monkeypatch):
    monkeypatch.setenv("RCMARL_LAT_W8", w8)
    KC.check_lattice_sgd_fit(bk, 1, 7, 130, 2, 5, 5, steps=2, masked_agent=2)


ENC_CASE, FWD_CASE, FIT_CASE, FIT_STEPS = (1, 7, 33, 3, 9, 9, True), (2, 5, 70, 2, 5, 5), (1, 7, 130, 3, 16, 16, 2), 2
@pytest.mark.parametrize("mode", ["0", "1", "3"])
def test_lattice_operand_forms(bk, mode, monkeypatch, lattice_form):
    """RCMARL_LAT_F16: 0 = three exact bf16 pieces everywhere, 1 = the forward operand as two f16 pieces of 2^10 alpha W1,
    3 (default) = the backward operand (2^8 dz1) too -- encode images, piece reconstruction, forward vs float64, whole SGD fits vs
    the oracle, in every form."""
    lattice_form(bk, mode)
    assert bk.lib.rcmarl_lattice_f16_mode() == int(mode)
    KC.check_lattice_encode(bk, *ENC_CASE)
    KC.check_lattice_forward(bk, *FWD_CASE)
    KC.check_lattice_sgd_fit(bk, *FIT_CASE[:-1], steps=FIT_STEPS, masked_agent=FIT_CASE[-1])


def test_lattice_f16_pieces_saturate_instead_of_overflowing(bk, monkeypatch, lattice_form):
    lattice_form(bk, "3")
    KC.check_lattice_f16_saturation(bk)


def test_lattice_f16_pieces_in_the_subnormal_range(bk, monkeypatch, lattice_form):
    """Weights of 1e-6: every f16 piece of 2^10 alpha W1 is a subnormal (multiples of 2^-24).  The matrix core must not flush
    them (the result would be zero); what is lost is the form's stated absolute floor (2^-25 of the scaled unit)."""
    lattice_form(bk, "3")
    KC.check_lattice_forward(bk, *FWD_CASE, w_scale=1e-6, tol=2e-3)


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



def test_mid_fit_fp32_form_behind_the_f16_operand(bk, monkeypatch):
    """RCMARL_MIDFIT=5: rcmarl_mid_fit_lattice on k_mid_fit_v5 alone (f32-input MFMAs, fmaf-chain arithmetic) instead of the default
    k_mid_fit_v8 (f16 matrix core) + fix-up, emitting the same two-piece f16 operand -- against the same oracle fits."""
    monkeypatch.setenv("RCMARL_MIDFIT", "5")
    KC.check_lattice_sgd_fit(bk, 1, 5, 300, 2, 5, 5, steps=2, masked_agent=2)



def test_lattice_operand_form_mismatch_is_refused(bk, lattice_form):
    """A packed buffer remembers the operand form it was written in: switching the form between producer (rcmarl_w1_split,
    rcmarl_lattice_encode, rcmarl_mid_fit_lattice) and consumer (the two lattice GEMMs) is an RCMARL_ERR_ARG, not a garbage result."""
    KC.check_lattice_form_mismatch(bk, lattice_form)
