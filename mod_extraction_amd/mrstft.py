"""``mrstft`` loss of mod_extraction/losses.py:155-156 (auraloss.freq.MultiResolutionSTFTLoss with its
default arguments) as a ``torch.autograd.Function`` over the ``mx_mrstft_loss`` HIP kernels (forward
value + gradient w.r.t. the prediction in one call; the target gets no gradient, as in training)."""
import ctypes
import math
from typing import Sequence, Tuple

import torch
from torch import Tensor as T, nn

from . import _hip

MAXN = 2048


def _windows(fft_sizes: Sequence[int], win_lengths: Sequence[int], device) -> T:
    w = torch.zeros((len(fft_sizes), MAXN), dtype=torch.float32)
    for r, (n, wl) in enumerate(zip(fft_sizes, win_lengths)):
        left = (n - wl) // 2                               # torch.stft centres a short window in the frame
        w[r, left:left + wl] = torch.hann_window(wl)
    return w.to(device)


def scratch_floats(B: int, Tn: int, n_fft: int, hop: int) -> int:
    """Workspace floats of one resolution for the gradient (mirrors mr_ws_floats in csrc/mrstft.hip)."""
    frames = 1 + Tn // hop
    run = max(32, -(-n_fft // hop))
    run += run & 1
    runs = -(-frames // run)
    return 2 * B * (frames * hop + runs * max(n_fft - hop, 0))


def mrstft_value_and_grad(mod: "MultiResolutionSTFTLoss", a: T, t: T, need_grad: bool = True, scale: float = 1.0):
    """a, t: (B, T) rows (unit inner stride).  Returns (scale * loss as a device scalar, d (scale * loss) / d a or None).
    ``mod.last_terms`` afterwards holds the per-resolution terms of THIS call: [sc_0, logmag_0, ..., total] with the
    total (only) carrying ``scale`` -- the sc / logmag entries are the raw, unweighted terms."""
    assert a.shape == t.shape and a.ndim == 2 and a.stride(1) == 1 and t.stride(1) == 1
    B, Tn = a.shape
    dev = a.device
    n_res = len(mod.fft_sizes)
    frames = [1 + Tn // h for h in mod.hop_sizes]
    part = torch.empty(3 * B * max(-(-f // 8) for f in frames), device=dev, dtype=torch.float64)
    coef = torch.empty(n_res, device=dev, dtype=torch.float32)
    terms = torch.empty(2 * n_res + 1, device=dev, dtype=torch.float32)
    # the two time-domain gradient components of every resolution: per clip frames * hop run sums + one tail of n_fft - hop
    # positions per run of F frames (include/modex_hip.h, mx_mrstft_loss)
    scratch = torch.empty(sum(scratch_floats(B, Tn, n, h) for n, h in zip(mod.fft_sizes, mod.hop_sizes)), device=dev,
                          dtype=torch.float32) if need_grad else None
    dx = torch.empty((B, Tn), device=dev, dtype=torch.float32) if need_grad else None
    ffts = (ctypes.c_int32 * n_res)(*mod.fft_sizes)
    hops = (ctypes.c_int32 * n_res)(*mod.hop_sizes)
    win, tw = mod.buffers_on(dev)
    _hip.call("mx_mrstft_loss", a.data_ptr(), a.stride(0), t.data_ptr(), t.stride(0), B, Tn, n_res,
              ctypes.cast(ffts, ctypes.c_void_p), ctypes.cast(hops, ctypes.c_void_p), _hip.ptr(win), _hip.ptr(tw),
              float(mod.w_sc * scale), float(mod.w_log_mag * scale), float(mod.eps), _hip.ptr(part), _hip.ptr(coef),
              _hip.ptr(scratch), _hip.ptr(terms), _hip.ptr(dx), Tn, _hip.stream())
    mod.last_terms = terms.detach()
    return terms[2 * n_res], dx


class _MRSTFTFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y_hat: T, y: T, mod: "MultiResolutionSTFTLoss"):
        assert y_hat.shape == y.shape
        Tn = y_hat.size(-1)
        a = y_hat.reshape(-1, Tn).contiguous().float()
        t = y.reshape(-1, Tn).contiguous().float()
        need_grad = y_hat.requires_grad
        value, dx = mrstft_value_and_grad(mod, a, t, need_grad)
        ctx.save_for_backward(dx if need_grad else torch.empty(0, device=a.device))
        ctx.shape = y_hat.shape
        return value

    @staticmethod
    def backward(ctx, g: T):
        (dx,) = ctx.saved_tensors
        return (dx * g).view(ctx.shape), None, None


class MultiResolutionSTFTLoss(nn.Module):
    def __init__(self, fft_sizes: Sequence[int] = (1024, 2048, 512), hop_sizes: Sequence[int] = (120, 240, 50),
                 win_lengths: Sequence[int] = (600, 1200, 240), w_sc: float = 1.0, w_log_mag: float = 1.0,
                 eps: float = 1e-8) -> None:
        super().__init__()
        assert all(n in (512, 1024, 2048) for n in fft_sizes)
        self.fft_sizes, self.hop_sizes, self.win_lengths = list(fft_sizes), list(hop_sizes), list(win_lengths)
        self.w_sc, self.w_log_mag, self.eps = w_sc, w_log_mag, eps
        self._cache = None
        self.last_terms = None

    def buffers_on(self, device) -> Tuple[T, T]:
        if self._cache is None or self._cache[0].device != device:
            k = torch.arange(MAXN, dtype=torch.float64) * (-2.0 * math.pi / MAXN)
            tw = torch.stack([torch.cos(k), torch.sin(k)], dim=1).float().to(device)
            self._cache = (_windows(self.fft_sizes, self.win_lengths, device), tw)
        return self._cache

    def forward(self, input: T, target: T) -> T:
        return _MRSTFTFn.apply(input, target, self)
