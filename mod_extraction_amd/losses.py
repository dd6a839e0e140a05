"""Host mirror of mod_extraction/losses.py.

``get_loss_func_by_name`` (losses.py:142-160) returns ``nn.Module``s with the reference's names and
call signature; l1 / fdl1 / sdl1 / mse -- the LFO-extraction losses -- are evaluated by the fused
``mx_lfo_loss`` HIP kernel (all four terms and d/d(y_hat) in one launch).  ``lfo_loss`` is the fused
weighted form that ``lightning.LFOExtraction`` uses (lightning.py:33-62).
"""
from typing import Dict, Tuple

import torch
from torch import Tensor as T, nn

from . import _hip

_LFO_TERMS = ("l1", "fdl1", "sdl1", "mse")


class _LFOLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y_hat: T, y: T, w_l1: float, w_fd: float, w_sd: float, w_mse: float):
        assert y_hat.shape == y.shape
        n = y_hat.size(-1)
        yh = y_hat.reshape(-1, n).contiguous().float()
        yt = y.reshape(-1, n).contiguous().float()
        B = yh.size(0)
        part = torch.empty((B, 4), device=yh.device, dtype=torch.float32)
        losses = torch.empty(5, device=yh.device, dtype=torch.float32)
        grad = torch.empty_like(yh)
        _hip.call("mx_lfo_loss", _hip.ptr(yh), _hip.ptr(yt), B, n, float(w_l1), float(w_fd), float(w_sd),
                  float(w_mse), _hip.ptr(part), _hip.ptr(losses), _hip.ptr(grad), _hip.stream())
        ctx.save_for_backward(grad)
        ctx.shape = y_hat.shape
        return losses

    @staticmethod
    def backward(ctx, g: T):
        (grad,) = ctx.saved_tensors
        return (grad * g[4]).view(ctx.shape), None, None, None, None, None


def lfo_loss(y_hat: T, y: T, weights: Dict[str, float]) -> Tuple[T, Dict[str, T]]:
    """Weighted LFO loss: returns (total, {name: term}); only weights > 0 enter the total
    (lightning.py:48-52) but every named term is reported."""
    for k in weights:
        if k not in _LFO_TERMS:
            raise KeyError(k)
    # losses.py:12 (central_diff asserts more than 2 points): the reference raises for rows too short to differentiate
    n = y_hat.size(-1)
    assert "fdl1" not in weights or n > 2, "fdl1: central difference needs more than 2 points"
    assert "sdl1" not in weights or n > 4, "sdl1: second central difference needs more than 4 points"
    w = [float(weights.get(k, 0.0)) for k in _LFO_TERMS]
    losses = _LFOLossFn.apply(y_hat, y, *w)
    return losses[4], {k: losses[i].detach() for i, k in enumerate(_LFO_TERMS) if k in weights}


class _SingleTerm(nn.Module):
    term = "l1"

    def forward(self, input: T, target: T) -> T:
        return lfo_loss(input, target, {self.term: 1.0})[0]


class L1Loss(_SingleTerm):
    term = "l1"


class FirstDerivativeL1Loss(_SingleTerm):
    term = "fdl1"

    @staticmethod
    def calc_first_derivative(x: T) -> T:
        """losses.py:80-84 (analysis helper; the loss itself differentiates inside the kernel)."""
        assert x.size(-1) > 2
        return (x[..., 2:] - x[..., :-2]) / 2.0


class SecondDerivativeL1Loss(_SingleTerm):
    term = "sdl1"

    @staticmethod
    def calc_second_derivative(x: T) -> T:
        """losses.py:97-102."""
        return FirstDerivativeL1Loss.calc_first_derivative(FirstDerivativeL1Loss.calc_first_derivative(x))


class MSELoss(_SingleTerm):
    term = "mse"


class LogMelLoss(nn.Module):
    """losses.py:105-130 (``log_mel_l1``; no shipped config uses it): L1 between the log-mel spectrograms of input and
    target on the K4 kernel.  A METRIC here: the log-mel kernel has no backward, so a prediction that requires grad
    raises instead of silently returning a constant."""

    def __init__(self, sr: float = 44100, n_fft: int = 1024, hop_len: int = 256, n_mels: int = 256,
                 eps: float = 1e-7) -> None:
        super().__init__()
        from .models import MelSpectrogramHIP, PITCH
        self.eps, self.hop_len, self.pitch = eps, hop_len, PITCH
        self.spectrogram = MelSpectrogramHIP(int(sr), n_fft, hop_len, n_mels)

    def forward(self, input: T, target: T) -> T:
        if input.requires_grad and torch.is_grad_enabled():
            raise NotImplementedError("log_mel_l1 is forward-only on this path (evaluation metric)")
        assert input.shape == target.shape and input.ndim == 3
        n_frames = input.size(-1) // self.hop_len + 1
        if n_frames > self.pitch:
            raise NotImplementedError(f"log_mel_l1: at most {self.pitch} frames per clip")
        if self.spectrogram.mel_scale.fb.device != input.device:
            self.spectrogram.to(input.device)
        a = self.spectrogram.log_mel(input.detach(), n_frames, self.eps)[..., :n_frames]
        b = self.spectrogram.log_mel(target.detach(), n_frames, self.eps)[..., :n_frames]
        return (a - b).abs().mean()


def apply_reduction(losses: T, reduction: str = "none") -> T:
    """losses.py:133-139."""
    if reduction == "mean":
        return losses.mean()
    if reduction == "sum":
        return losses.sum()
    return losses


def get_loss_func_by_name(name: str) -> nn.Module:
    if name == "l1":
        return L1Loss()
    elif name == "fdl1":
        return FirstDerivativeL1Loss()
    elif name == "sdl1":
        return SecondDerivativeL1Loss()
    elif name == "mse":
        return MSELoss()
    elif name in ("esr", "dc"):
        from .effect_losses import get_effect_loss
        return get_effect_loss(name)
    elif name == "mrstft":
        from .effect_losses import get_effect_loss
        return get_effect_loss(name)
    elif name == "log_mel_l1":
        return LogMelLoss()
    else:
        raise KeyError
