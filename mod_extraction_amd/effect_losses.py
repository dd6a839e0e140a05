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
