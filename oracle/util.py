"""Oracle restatement of mod_extraction/util.py (TEST INFRASTRUCTURE ONLY).

* ``linear_interpolate_last_dim``  follows util.py:15-29, i.e. ``F.interpolate(mode="linear",
  align_corners=True)``.  Restated explicitly (not by calling F.interpolate) so the rounding
  sequence the HIP kernel must reproduce is written down:
      scale = fl32(in-1) / fl32(out-1);  real = scale * fl32(i)
      i0 = min(int(real), in-1);  lam1 = clamp(real - i0, 0, 1);  lam0 = 1 - lam1
      out = fma(lam0, x[i0], fl32(lam1 * x[i1]))          <- torch CPU contracts exactly this way
  (probed bit-exact against torch 2.10 CPU on 7 size pairs; pinned by tests/golden/interp.npz).
* RNG helpers follow util.py:32-62 (torch global generator; scipy loguniform -> numpy global RNG).
"""
from typing import Any, List, Union

import numpy as np
import torch


def interp_indices_weights(n_in: int, n_out: int):
    """(i0, i1, lam0, lam1) of aten's linear/align_corners=True source-index rule."""
    if n_out > 1:
        scale = np.float32(np.float32(n_in - 1) / np.float32(n_out - 1))
    else:
        scale = np.float32(0.0)
    real = (scale * np.arange(n_out, dtype=np.float32)).astype(np.float32)
    i0 = np.minimum(real.astype(np.int64), n_in - 1)
    lam1 = np.clip(real - i0.astype(np.float32), np.float32(0), np.float32(1)).astype(np.float32)
    i1 = i0 + (i0 < n_in - 1)
    lam0 = (np.float32(1.0) - lam1).astype(np.float32)
    return i0, i1, lam0, lam1


def linear_interpolate_last_dim_np(x: np.ndarray, n: int) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    if x.shape[-1] == n:
        return x
    i0, i1, lam0, lam1 = interp_indices_weights(x.shape[-1], n)
    second = (lam1 * x[..., i1]).astype(np.float32)
    # fused multiply-add emulated in float64 (exact for fp32 operands), rounded once
    out = lam0.astype(np.float64) * x[..., i0].astype(np.float64) + second.astype(np.float64)
    return out.astype(np.float32)


def linear_interpolate_last_dim(x: torch.Tensor, n: int, align_corners: bool = True) -> torch.Tensor:
    assert align_corners, "the reference only ever uses align_corners=True on this path"
    assert 1 <= x.ndim <= 3
    return torch.from_numpy(linear_interpolate_last_dim_np(x.detach().cpu().numpy(), n))


# ---- RNG helpers (util.py:32-62) -------------------------------------------------------------
def randint(low: int, high: int, n: int = 1) -> Union[int, torch.Tensor]:
    v = torch.randint(low=low, high=high, size=(n,))
    return v.item() if n == 1 else v


def choice(items: List[Any]) -> Any:
    assert len(items) > 0
    return items[randint(0, len(items))]


def sample_uniform(low: float, high: float, n: int = 1) -> Union[float, torch.Tensor]:
    v = torch.rand(n) * (high - low) + low
    return v.item() if n == 1 else v


def sample_log_uniform(low: float, high: float, n: int = 1) -> Union[float, torch.Tensor]:
    from scipy.stats import loguniform
    if low == high:
        return low if n == 1 else torch.full(size=(n,), fill_value=low)
    v = loguniform.rvs(low, high, size=n)
    return float(v) if n == 1 else torch.from_numpy(v)
