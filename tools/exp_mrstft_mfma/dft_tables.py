"""Host-side tables of the matrix-core DFT experiment (tools/exp_mrstft_mfma/README.md): fragments of the 32- and
16-point DFT matrices as fp16 pairs in MFMA operand order, and the per-lane twiddles."""
import torch
T = torch.Tensor

# ---- constant tables of the matrix-core DFT (layout: csrc/mrstft.hip, MR_T_* offsets) ----------------------------------
T_FLOATS = 12288
_SF = 1024.0                      # scale of the constant fragments


def _row(r: int, hh: int) -> int:
    """row of accumulator register r in lane half hh of v_mfma_f32_32x32x16_f16 = index of operand slot (q, j) = (r >> 3, r & 7)"""
    return (r & 3) + 8 * (r >> 2) + 4 * hh


def _fragments(entry) -> "np.ndarray":
    """halfs [re_hi, re_lo, im_hi, im_lo][q][lane][8] of the complex 32 x 32 constant entry(idx, slot_index) * 1024, as the
    fp16 pair hi = fp16(v), lo = fp16(v - hi); returned as the float32 view of the 4096 halfs."""
    import numpy as np
    out = np.zeros((4, 2, 64, 8), dtype=np.float16)
    for q in range(2):
        for lane in range(64):
            idx, hh = lane & 31, lane >> 5
            for j in range(8):
                v = complex(entry(idx, _row(8 * q + j, hh))) * _SF
                for part, val in ((0, v.real), (2, v.imag)):
                    hi = np.float16(np.float32(val))
                    out[part, q, lane, j] = hi
                    out[part + 1, q, lane, j] = np.float16(np.float32(val) - np.float32(hi))
    return out.reshape(-1).view(np.float32)


def dft_tables() -> T:
    """The T_FLOATS floats mx_mrstft_loss takes as `dft_tables`, evaluated in fp64 on the host:
    fragments of F32[a][b] = exp(-2 pi i a b / 32) and of blockdiag(F16, F16), and the per-lane twiddles [r][lane]."""
    import numpy as np
    w = lambda n, e: np.exp(-2j * np.pi * (e % n) / n)
    tab = np.zeros(T_FLOATS, dtype=np.float32)
    tab[0:2048] = _fragments(lambda idx, slot: w(32, idx * slot))
    tab[2048:4096] = _fragments(lambda idx, slot: w(16, (idx & 15) * (slot & 15)) if (idx >> 4) == (slot >> 4) else 0.0)

    def twiddles(exponent, n, scale=1.0):
        t = np.zeros((16, 64, 2), dtype=np.float32)
        for r in range(16):
            for lane in range(64):
                v = w(n, exponent(_row(r, lane >> 5), lane & 31)) * scale
                t[r, lane] = (v.real, v.imag)
        return t.reshape(-1)
    between = 1.0 / (32.0 * _SF)       # the twiddles between the two steps also carry the rescaling of the accumulator (2^-15)
    tab[4096:6144] = twiddles(lambda row, idx: row * idx, 1024, between)
    tab[6144:8192] = twiddles(lambda row, idx: (row & 15) * idx, 512, between)
    tab[8192:10240] = twiddles(lambda row, idx: row * (idx & 15), 512, between)
    tab[10240:12288] = twiddles(lambda row, idx: idx + 32 * row, 2048)
    return torch.from_numpy(tab)


