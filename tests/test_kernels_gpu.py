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
        res[mx] = KC.check_consensus_head(bk, 2, 5, 1000, 10, 4, 1, "circ", outlier=1e4, compare=False)
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
@pytest.mark.parametrize("S,N,B,in_dim,row_off", [(2, 3, 300, 6, 0), (1, 5, 700, 10, 1), (1, 2, 64, 4, 0), (2, 64, 3000, 128, 1)])
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


@pytest.mark.parametrize("S,N,B,in_dim", [(2, 5, 1000, 10), (1, 16, 300, 32), (1, 64, 1000, 128), (1, 256, 1000, 512)])     # last: BASELINE configs[3]
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
    """default: <= 20 inputs on the f16 matrix core (k_minibatch_mx + fp32 fix-up for out-of-range networks); one network of the
    longest chains may sit on a LeakyReLU knife-edge (kernel_checks.check_minibatch_fit)"""
    KC.check_minibatch_fit(bk, S, N, B, in_dim, advs, bs=bs, epochs=3, shuffle=shuffle, knife_edge_nets=1 if (in_dim <= 20 and B >= 900) else 0)


@pytest.mark.parametrize("S,N,B,in_dim,advs,bs,shuffle", [(2, 5, 1000, 10, [4], 32, True), (1, 5, 3000, 15, [1, 3], 32, True),
                                                          (9, 5, 333, 20, [0], 7, True)])
def test_minibatch_fit_fp32_wavefront_kernel(bk, S, N, B, in_dim, advs, bs, shuffle, monkeypatch):
    """RCMARL_MB_MX=0: k_minibatch_wave alone (the oracle's fmaf chains: every network at the strict bar)"""
    monkeypatch.setenv("RCMARL_MB_MX", "0")
    KC.check_minibatch_fit(bk, S, N, B, in_dim, advs, bs=bs, epochs=3, shuffle=shuffle)


@pytest.mark.parametrize("S,N,B,in_dim,advs,bs,shuffle", [(2, 5, 1000, 10, [4], 32, True), (9, 5, 333, 20, [0], 7, True), (300, 5, 400, 15, [0, 1, 2, 3, 4], 400, False), (7, 6, 900, 18, [2, 5], 900, False)])
def test_minibatch_fit_compact_form(bk, S, N, B, in_dim, advs, bs, shuffle, monkeypatch):
    """RCMARL_MB_MX_COMPACT=1: k_minibatch_mx with the short k-step-1 weight fragments and ONE B plane pair for both gradient
    products (eight wavefronts per CU; the default from 1537 networks per launch on), mini-batches and full batches (bs = B:
    the cooperative agents' local fit, engine._fit_as_chains)."""
    monkeypatch.setenv("RCMARL_MB_MX_COMPACT", "1")
    KC.check_minibatch_fit(bk, S, N, B, in_dim, advs, bs=bs, epochs=3, shuffle=shuffle, knife_edge_nets=1 if B >= 900 else 0)


@pytest.mark.parametrize("seeds,calls,epochs,B", [((3, 77, 1000, 12345678901), (0, 5, 29), 10, 3000), ((11,), (2,), 1, 1), (tuple(range(64)), (0, 1, 2), 3, 777), ((5,), (7,), 2, 8192)])
def test_shuffle_perms_equal_the_oracle_stream(bk, seeds, calls, epochs, B):
    KC.check_shuffle_perms(bk, seeds, calls, epochs, B)


def test_minibatch_fit_out_of_range_network_is_redone_in_fp32(bk, monkeypatch):
    """A network whose weights leave the f16 range (2^10 |W| > 65000) is flagged by k_minibatch_mx, not written back, and redone by
    the fp32 kernel in the fix-up launch: its result equals RCMARL_MB_MX=0 bit for bit; the healthy network beside it is untouched
    by the fix-up (its result differs from the fp32 kernel's in the last bits only)."""
    import numpy as np
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("RCMARL_MB_MX", mode)
        res[mode] = KC.run_minibatch_fit_with_blown_network(bk)
    np.testing.assert_array_equal(res["1"][0], res["0"][0])                       # the blown-up network: the fp32 kernel's bits
    assert not np.array_equal(res["1"][1], res["0"][1])                           # the healthy one did run on the matrix core ...
    assert np.abs(res["1"][1] - res["0"][1]).max() <= 1e-5                        # ... and agrees with the fp32 kernel


@pytest.mark.parametrize("S,N,B,in_dim,advs,bs,t0", [(2, 5, 1000, 10, [4], 200, 0), (1, 5, 1000, 10, [0, 2], 200, 15),
                                                     (1, 64, 400, 128, [5], 200, 3)])
def test_minibatch_actor(bk, S, N, B, in_dim, advs, bs, t0):
    KC.check_minibatch_actor(bk, S, N, B, in_dim, advs, bs=bs, t0=t0, shuffle=True)


@pytest.mark.parametrize("in_dims,blow,compact", [((10, 15, 10), None, None), ((18, 20, 18), None, None),      # both input classes
                                                  ((10, 15, 10), (1, 0), None), ((18, 20, 18), (2, 1), "1")])    # a flag set by job j > 0
def test_minibatch_fit_multi_equals_single_job_launches(bk, in_dims, blow, compact, monkeypatch):
    if compact is not None:
        monkeypatch.setenv("RCMARL_MB_MX_COMPACT", compact)
    KC.check_minibatch_fit_multi(bk, in_dims=in_dims, blow=blow)


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


@pytest.mark.parametrize("S,N,B,in_dim", [(2, 40, 700, 80),            # two workgroups per agent
                                          (16, 128, 1500, 256),        # 2048 columns: one workgroup walks all six chunks of an agent
                                          (16, 256, 3000, 512)])       # BASELINE configs[3]
def test_mid_step_f16_kernel_vs_fp32_kernel(bk, S, N, B, in_dim):
    KC.check_mid_step_f16_vs_fp32_kernel(bk, S, N, B, in_dim)


@pytest.mark.parametrize("width", [2, 3])
def test_lattice_equals_f32_path_at_full_size(bk, width):
    """BASELINE configs[3] sizes (N=256 agents, B=3000 rows, 32x32 grid): the bf16x3 lattice GEMMs and the
    f32-MFMA GEMMs are independent implementations of the same fp32 math -> they must agree to fp32 roundoff
    (forward activations, and W1 after two full SGD steps through mid_fit)."""
    KC.check_lattice_vs_f32(bk, S=2, N=256, B=3000, width=width, nrow=32, ncol=32, steps=2)


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
@pytest.mark.parametrize("f16", [1, 0])
def test_wide_forward(bk, S, N, B, in_dim, hid, f16, wide_form):
    wide_form(bk, f16)
    WC.check_wide_forward(bk, S, N, B, in_dim, hid)


@pytest.mark.parametrize("S,N,B,in_dim,hid,masked", [(2, 5, 1000, 12, 64, None), (1, 3, 701, 10, 24, 1), (1, 4, 3000, 128, 512, 2),
                                                     (1, 16, 1000, 32, 128, None)])
@pytest.mark.parametrize("f16", [1, 0])
def test_wide_fit(bk, S, N, B, in_dim, hid, masked, f16, wide_form):
    wide_form(bk, f16)
    WC.check_wide_fit(bk, S, N, B, in_dim, hid, steps=3, masked_agent=masked)


@pytest.mark.parametrize("S,N,B,hid", [(2, 3, 70, 24), (1, 16, 3000, 512), (2, 37, 1000, 40)])
def test_pack_dz_with_row_sums(bk, S, N, B, hid):
    KC.check_pack_dz_rowsum(bk, S, N, B, hid)


def test_wide_dense_layer_out_of_f16_range_recomputes_in_fp32(bk):
    WC.check_wide_out_of_range(bk, *(1, 3, 700, 64, 256))


@pytest.mark.parametrize("S,N,B,in_dim,hid,d,H,graph", [(2, 5, 1000, 10, 64, 4, 1, "circ"), (1, 24, 700, 48, 128, 10, 4, "rand"),
                                                        (1, 70, 130, 140, 512, 66, 32, "circ"), (1, 30, 300, 60, 64, 23, 5, "rand")])
@pytest.mark.parametrize("f16", [1, 0])
def test_wide_consensus_head(bk, S, N, B, in_dim, hid, d, H, graph, f16, wide_form):
    wide_form(bk, f16)
    WC.check_wide_consensus_head(bk, S, N, B, in_dim, hid, d, H, graph)


def test_copy3d_more_batches_than_a_grid_dimension(bk):
    """rcmarl_copy3d with more than 65535 batches (AdversaryPath publishes message rows with batches = seeds): the grid's z extent is
    capped and the kernel strides over the rest; masked rows stay untouched."""
    rng = np.random.default_rng(9)
    batches, rows, cols, ld = 70001, 3, 8, 12
    src = rng.normal(size=(batches, rows, ld)).astype(np.float32)
    dst0 = rng.normal(size=(batches, rows, ld)).astype(np.float32)
    mask = np.array([1, 0, 1], np.int32)
    d_src, d_dst, d_mask = bk.dev(src), bk.dev(dst0), bk.dev(mask)
    bk.lib.rcmarl_copy3d(bk.ptr(d_src), rows * ld, ld, bk.ptr(d_dst), rows * ld, ld, batches, rows, cols, bk.ptr(d_mask), bk.stream)
    got = bk.host(d_dst)
    want = dst0.copy()
    want[:, [0, 2], :cols] = src[:, [0, 2], :cols]
    np.testing.assert_array_equal(got, want)


# ---- the same layers on pre-split packed operands (hid % 128 == 0): csrc/dense_pk.hip ---------------------------------
@pytest.mark.parametrize("S,N,B,width,nrow,ncol,hid", [(1, 8, 1000, 2, 16, 16, 512), (2, 5, 333, 3, 7, 9, 128), (1, 3, 3000, 3, 32, 32, 256)])
def test_pk_forward(bk, S, N, B, width, nrow, ncol, hid):
    WC.check_pk_forward(bk, S, N, B, width, nrow, ncol, hid)


@pytest.mark.parametrize("S,N,B,width,nrow,ncol,hid,steps,masked", [(1, 8, 1000, 2, 16, 16, 512, 3, None), (2, 16, 777, 3, 7, 9, 128, 3, 4),
                                                                    (1, 3, 3000, 3, 32, 32, 256, 5, 1), (3, 8, 130, 2, 5, 5, 128, 2, None)])
def test_pk_fit(bk, S, N, B, width, nrow, ncol, hid, steps, masked):
    worst = WC.check_pk_fit(bk, S, N, B, width, nrow, ncol, hid, steps=steps, lr=0.02, masked_agent=masked, tol=1e-5)
    print("packed-operand fit: worst |w - w_oracle| / max(1, |w|max) = %.2e" % worst)


def test_pk_operands_beyond_the_f16_range_raise_the_flag(bk):
    WC.check_pk_range_flag(bk)


@pytest.mark.parametrize("m128", ["0", "1"])
def test_lattice_backward_both_tile_heights(bk, m128, monkeypatch):
    """Networks of at most 128 inputs take 128-row tiles in the backward GEMM (RCMARL_LAT_M128=0: the 256-row tile of the wide
    inputs): the same products in the same k order, the same oracle fit either way."""
    monkeypatch.setenv("RCMARL_LAT_M128", m128)
    KC.check_lattice_sgd_fit(bk, 2, 40, 777, 3, 7, 9, steps=3, masked_agent=4)


@pytest.mark.parametrize("w8", ["0", "1"])
def test_lattice_gemms_both_wavefront_shapes(bk, w8, monkeypatch):
    """RCMARL_LAT_W8: the lattice GEMMs as four wavefronts of 64x128 / 128x64 or eight of 64x64 per workgroup -- same
    products, same k order: the same oracle fit either way (forward defaults to eight, backward to four)."""
    monkeypatch.setenv("RCMARL_LAT_W8", w8)
    KC.check_lattice_sgd_fit(bk, 2, 20, 777, 3, 7, 9, steps=3, masked_agent=4)
    KC.check_lattice_forward(bk, 1, 64, 1000, 2, 16, 16)


ENC_CASE, FWD_CASE, FIT_CASE, FIT_STEPS = (1, 64, 700, 3, 16, 16, True), (2, 64, 1000, 2, 16, 16), (2, 20, 777, 3, 7, 9, 4), 5
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
    for _ in range(6):
        N = int(rng.integers(d, 300))
        P_hid = int(rng.integers(1, 700))
        graph = "circ" if rng.random() < 0.7 else "rand"
        KC.check_consensus_params_exact(bk, N, d, H, P_hid, int(rng.integers(1, 3)), int(rng.integers(1 << 30)), graph)



@pytest.mark.parametrize("S,N,B,width,nrow,ncol,masked", [(2, 5, 1000, 2, 5, 5, None), (1, 64, 1000, 3, 16, 16, 5), (1, 20, 777, 2, 7, 9, 3)])
def test_mid_fit_fp32_form_behind_the_f16_operand(bk, S, N, B, width, nrow, ncol, masked, monkeypatch):
    """RCMARL_MIDFIT=5: rcmarl_mid_fit_lattice on k_mid_fit_v5 alone (f32-input MFMAs, fmaf-chain arithmetic) instead of the default
    k_mid_fit_v8 (f16 matrix core) + fix-up, emitting the same two-piece f16 operand -- against the same oracle fits."""
    monkeypatch.setenv("RCMARL_MIDFIT", "5")
    KC.check_lattice_sgd_fit(bk, S, N, B, width, nrow, ncol, steps=5, masked_agent=masked)



def test_lattice_operand_form_mismatch_is_refused(bk, lattice_form):
    """A packed buffer remembers the operand form it was written in: switching the form between producer (rcmarl_w1_split,
    rcmarl_lattice_encode, rcmarl_mid_fit_lattice) and consumer (the two lattice GEMMs) is an RCMARL_ERR_ARG, not a garbage result."""
    KC.check_lattice_form_mismatch(bk, lattice_form)
