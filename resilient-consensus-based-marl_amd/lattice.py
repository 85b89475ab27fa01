"""Host-side helpers of the lattice (exact bf16x3) layer-1 path: buffer geometry of the
packed bf16 operands and the per-column lattice constants.  The arithmetic argument and the
byte layout live in csrc/rcmarl_lattice.h; the kernels in csrc/lattice_gemm.hip.

Why the path exists: every network input column of the reference is either a z-scored grid
coordinate ``(pos - mean)/std`` with ``mean = (n-1)/2`` (environments/grid_world.py:29-33,66-72)
or a raw action index (training/train_agents.py:91), i.e. ``alpha_c * K`` with a small integer
K.  Integers up to 256 are exact in bf16 and a fp32 value splits exactly into three bf16 pieces,
so the two layer-1 GEMMs run on the bf16 matrix core with exact products and fp32 accumulation.
"""
import numpy as np

PK_BLOCK = 8192          # bytes of one [128 rows][32 k] bf16 block
HID = 20


def cdiv(a, b):
    return (int(a) + b - 1) // b


class Geometry:
    """Allocated tile counts (rt = 128-row tiles, kt = 32-deep k-tiles) of the four packed operands
    for N agents, input width in_dim and at most `cap` replay rows."""

    def __init__(self, n_agents, in_dim, cap, hid=HID):
        self.N, self.in_dim, self.cap, self.hid = int(n_agents), int(in_dim), int(cap), int(hid)
        b_pad = cdiv(cap, 256) * 256
        self.kp = (b_pad // 128, cdiv(in_dim, 32))                    # rows = replay row, k = feature
        self.ktp = (2 * cdiv(in_dim, 256), b_pad // 32)               # rows = feature, k = replay row
        self.wp = (cdiv(n_agents * hid, 128), cdiv(in_dim, 32))       # rows = (agent,unit), k = feature, 3 pieces
        self.dzp = (cdiv(n_agents * hid, 128), b_pad // 32)           # rows = (agent,unit), k = replay row, 3 pieces

    @staticmethod
    def nbytes(rt_kt, pieces):
        return rt_kt[0] * rt_kt[1] * pieces * PK_BLOCK


def column_alpha(n_agents, width, nrow, ncol, scaling):
    """alpha_c (fp32) per input column for a state (width 2: x,y per agent) or state-action
    (width 3: x,y,a per agent) row: x = alpha * K with K = 2*pos-(n-1) (scaled) or pos (unscaled)."""
    if scaling:
        ax = np.float32(0.5 / np.std(np.arange(nrow)))
        ay = np.float32(0.5 / np.std(np.arange(ncol)))
    else:
        ax = ay = np.float32(1.0)
    per_agent = [ax, ay] + ([np.float32(1.0)] if width == 3 else [])
    return np.tile(np.asarray(per_agent, np.float32), n_agents)


def pk_element_index(n_rows, n_k, kt_alloc, pieces, piece=0):
    """uint16-element index [n_rows][n_k] of (row, k) of `piece` inside one seed's packed buffer."""
    r = np.arange(n_rows, dtype=np.int64)[:, None]
    k = np.arange(n_k, dtype=np.int64)[None, :]
    off = (((r >> 7) * kt_alloc + (k >> 5)) * pieces + piece) * PK_BLOCK + (r & 127) * 64 + \
        ((((k & 31) >> 3) ^ ((r >> 2) & 3)) << 4) + (k & 7) * 2
    return off // 2


def bf16_bits_to_f32(u16):
    return (np.asarray(u16, np.uint32) << 16).view(np.float32)


def f16_bits_to_f32(u16):
    return np.asarray(u16, np.uint16).view(np.float16).astype(np.float32)


def pk_unpack(buf_u16, n_rows, n_k, kt_alloc, pieces, f16=False):
    """Decode one seed's packed buffer (1-D uint16 view) -> float32 [pieces][n_rows][n_k] (bf16 pieces, or f16 ones)."""
    dec = f16_bits_to_f32 if f16 else bf16_bits_to_f32
    return np.stack([dec(buf_u16[pk_element_index(n_rows, n_k, kt_alloc, pieces, p)]) for p in range(pieces)])


# the two-piece f16 form (csrc/rcmarl_lattice.h): fixed scales of the forward / backward operand
F16_W_SCALE, F16_DZ_SCALE = 1024.0, 256.0
