"""Host mirror of mod_extraction/util.py on HIP tensors.

``linear_interpolate_last_dim`` (util.py:15-29) runs the ``mx_interp_linear`` kernel; the RNG helpers
(util.py:32-62) stay on the host exactly as in the reference (torch global generator, scipy
``loguniform`` drawing from numpy's global RNG) so seeded parameter streams are identical.
"""
from typing import Any, List, Union

import torch
from torch import Tensor as T

from . import _hip


def linear_interpolate_last_dim(x: T, n: int, align_corners: bool = True) -> T:
    if not align_corners:
        raise NotImplementedError("the reference only uses align_corners=True on this path")
    assert 1 <= x.ndim <= 3
    if x.size(-1) == n:
        return x
    xc = x.contiguous().float()
    rows = xc.numel() // xc.size(-1)
    y = torch.empty(xc.shape[:-1] + (n,), device=xc.device, dtype=torch.float32)
    _hip.call("mx_interp_linear", _hip.ptr(xc), rows, xc.size(-1), n, _hip.ptr(y), _hip.stream())
    return y


def linear_interpolate_last_dim_bwd(dy: T, n_in: int, n_out: int, j0: int = 0) -> T:
    """Transpose of ``linear_interpolate_last_dim`` for a window of the output axis: ``dy`` (..., j_len) is the gradient
    w.r.t. ``y[..., j0 : j0 + j_len]`` (zero elsewhere); returns the gradient w.r.t. the (..., n_in) input."""
    assert 1 <= dy.ndim <= 3 and dy.stride(-1) == 1
    lead = dy.shape[:-1]
    d2 = dy.reshape(-1, dy.size(-1)).contiguous().float()
    dx = torch.empty((d2.size(0), n_in), device=dy.device, dtype=torch.float32)
    _hip.call("mx_interp_linear_bwd", _hip.ptr(d2), d2.stride(0), d2.size(0), n_in, n_out, j0, d2.size(1), _hip.ptr(dx),
              _hip.stream())
    return dx.view(lead + (n_in,))


def randint(low: int, high: int, n: int = 1) -> Union[int, T]:
    x = torch.randint(low=low, high=high, size=(n,))
    return x.item() if n == 1 else x


def choice(items: List[Any]) -> Any:
    assert len(items) > 0
    return items[randint(0, len(items))]


def sample_uniform(low: float, high: float, n: int = 1) -> Union[float, T]:
    x = (torch.rand(n) * (high - low)) + low
    return x.item() if n == 1 else x


def sample_log_uniform(low: float, high: float, n: int = 1) -> Union[float, T]:
    from scipy.stats import loguniform
    if low == high:
        return low if n == 1 else torch.full(size=(n,), fill_value=low)
    x = loguniform.rvs(low, high, size=n)
    return float(x[0]) if n == 1 else torch.from_numpy(x)
