"""Host mirror of mod_extraction/modulations.py on HIP tensors.

``make_mod_signal`` keeps the reference signature (modulations.py:16-21) and runs the ``mx_lfo_synth``
kernel; ``make_mod_signals`` is the batched form the training loop uses (one launch for the whole
batch, optional crop offset and on-the-fly resampling).
"""
import math
from typing import List, Optional, Sequence, Union

import torch
from torch import Tensor as T

from . import _hip

SHAPE_IDS = {"cos": 0, "rect_cos": 1, "inv_rect_cos": 2, "tri": 3, "saw": 4, "rsaw": 5, "sqr": 6}


def _device(device=None) -> torch.device:
    return torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())


def make_mod_signals(n_samples: int, sr: float, freq: T, phase: T, shape: Optional[T] = None,
                     exp: Optional[T] = None, start: Optional[T] = None,
                     n_out: Optional[int] = None, out: Optional[T] = None) -> T:
    """Batched LFO synthesis: freq/phase/exp (B,) fp32, shape/start (B,) int32, all on the device.
    Returns (B, n_out or n_samples)."""
    B = freq.numel()
    n_out = n_samples if n_out is None else n_out
    y = out if out is not None else torch.empty((B, n_out), device=freq.device, dtype=torch.float32)
    _hip.call("mx_lfo_synth", _hip.ptr(freq), _hip.ptr(phase), _hip.ptr(shape), _hip.ptr(exp),
              _hip.ptr(start), B, n_samples, n_out, float(sr), _hip.ptr(y), _hip.stream())
    return y


def make_mod_signal(n_samples: int, sr: float, freq: float, phase: float = 0.0, shape: str = "cos",
                    exp: float = 1.0, device=None) -> T:
    assert n_samples > 0
    assert 0.0 < freq < sr / 2.0
    assert -2 * math.pi <= phase <= 2 * math.pi
    assert shape in SHAPE_IDS
    assert exp > 0
    dev = _device(device)
    f = torch.tensor([float(freq)], device=dev, dtype=torch.float32)
    p = torch.tensor([float(phase)], device=dev, dtype=torch.float32)
    s = torch.tensor([SHAPE_IDS[shape]], device=dev, dtype=torch.int32)
    e = torch.tensor([float(exp)], device=dev, dtype=torch.float32)
    return make_mod_signals(n_samples, sr, f, p, s, e).view(-1)
