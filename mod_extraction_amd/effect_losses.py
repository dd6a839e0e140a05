"""Effect-model losses of mod_extraction/losses.py:14-67 (ESR, DC) and nn.L1Loss on HIP tensors:
one ``mx_effect_loss_sums`` launch gives the per-clip sums all three are built from."""
from typing import Dict

import torch
from torch import Tensor as T, nn

from . import _hip


def effect_loss_terms(y_hat: T, y: T, eps: float = 1e-8) -> Dict[str, T]:
    """y_hat, y: (B,1,T) device tensors -> {"l1", "esr", "dc", "mse"} scalars ('mean' reductions)."""
    assert y_hat.shape == y.shape and y_hat.ndim == 3 and y_hat.size(1) == 1
    a, t = y_hat.detach()[:, 0, :], y.detach()[:, 0, :]
    assert a.stride(1) == 1 and t.stride(1) == 1
    B, Tn = a.shape
    part = torch.empty((B, 4), device=a.device, dtype=torch.float32)
    _hip.call("mx_effect_loss_sums", a.data_ptr(), a.stride(0), t.data_ptr(), t.stride(0), B, Tn, _hip.ptr(part),
              _hip.stream())
    s_abs, s_sq, s_yy, s_e = part[:, 0], part[:, 1], part[:, 2], part[:, 3]
    return {"l1": s_abs.sum() / (B * Tn), "mse": s_sq.sum() / (B * Tn),
            "esr": (s_sq / (s_yy + eps)).mean(),                                  # losses.py:33-38
            "dc": ((s_e / Tn) ** 2 / (s_yy / Tn + eps)).mean()}                   # losses.py:61-66


GRAD_NAMES = ("l1", "mse", "esr", "dc", "mrstft")      # losses whose d/dy_hat the TBPTT step can back-propagate


def effect_loss_grad(y_hat: T, y: T, weights: Dict[str, float], eps: float = 1e-8, mrstft=None) -> T:
    """d (sum_k weights[k] * loss_k(y_hat, y)) / d y_hat as a (B, T) tensor -- the backward half of
    ``calc_and_log_losses`` (lightning.py:33-62,380-382) for the effect model's output chunk.  ``mrstft``: a
    ``MultiResolutionSTFTLoss`` module to reuse (window / twiddle tables)."""
    assert y_hat.shape == y.shape and y_hat.ndim == 3 and y_hat.size(1) == 1
    a, t = y_hat.detach()[:, 0, :], y.detach()[:, 0, :]
    assert a.stride(1) == 1 and t.stride(1) == 1
    B, Tn = a.shape
    w = {k: float(v) for k, v in weights.items() if v > 0}
    unknown = [k for k in w if k not in GRAD_NAMES]
    if unknown:
        raise NotImplementedError(f"effect-model loss(es) {unknown} have no gradient kernel (supported: {GRAD_NAMES})")
    dy, acc = None, 0
    if "mrstft" in w:
        from .mrstft import MultiResolutionSTFTLoss, mrstft_value_and_grad
        mod = mrstft if mrstft is not None else MultiResolutionSTFTLoss()
        _, dy = mrstft_value_and_grad(mod, a, t, scale=w["mrstft"])
        acc = 1
    if dy is None:
        dy = torch.empty((B, Tn), device=a.device, dtype=torch.float32)
    if acc == 0 or any(k in w for k in ("l1", "mse", "esr", "dc")):
        _hip.call("mx_effect_loss_grad", a.data_ptr(), a.stride(0), t.data_ptr(), t.stride(0), B, Tn, w.get("l1", 0.0),
                  w.get("mse", 0.0), w.get("esr", 0.0), w.get("dc", 0.0), float(eps), acc, _hip.ptr(dy), dy.stride(0),
                  _hip.stream())
    return dy


class _Term(nn.Module):
    name = "l1"

    def forward(self, input: T, target: T) -> T:
        return effect_loss_terms(input, target)[self.name]


class ESRLoss(_Term):
    name = "esr"


class DCLoss(_Term):
    name = "dc"


def get_effect_loss(name: str) -> nn.Module:
    if name == "esr":
        return ESRLoss()
    if name == "dc":
        return DCLoss()
    if name == "mrstft":
        from .mrstft import MultiResolutionSTFTLoss
        return MultiResolutionSTFTLoss()
    raise KeyError(name)
