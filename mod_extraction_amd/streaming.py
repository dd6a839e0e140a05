"""Streaming inference of the LSTM effect model: host mirror of `EffectModel` and of
the parameter mapping of `EffectModelWrapper.do_forward_pass` in scripts/export_neutone_models.py:20-49,117-124.

A stereo buffer (2, n) is one batch of two clips; the conditioning LFO is generated on the fly,
    arg_l[k] = cumsum(2 pi f / sr)[k] + prev_phase,   arg_r = arg_l + stereo offset,   lfo = depth (cos(arg) + 1) / 2,
with `prev_phase = arg_l[-1] mod 2 pi` carried to the next buffer, and the LSTM-64 (`mx_lstm_fwd`) carries its
hidden state across buffers.  torch's CPU cumsum of a constant fp32 step equals fl32(fl64(k+1) * fl64(step)), which
is how the argument is formed here (same closed form as `mx_lfo_synth`), so the phase bookkeeping is bit-identical
to the reference's; cos runs on the device (1 ulp).  The Neutone packaging itself (TorchScript export, metadata) is
out of scope.
"""
import math
import os
from typing import Dict, Optional

import torch
from torch import Tensor as T
from torch import nn

from .models import LSTMEffectModel


class EffectModel(nn.Module):
    def __init__(self, weights_path: Optional[str] = None, n_hidden: int = 64, sr: float = 44100) -> None:
        super().__init__()
        self.sr = sr
        self.model = LSTMEffectModel(in_ch=1, out_ch=1, n_hidden=n_hidden, latent_dim=1)
        if weights_path:
            assert os.path.isfile(weights_path)
            self.model.load_state_dict(torch.load(weights_path, map_location="cpu"))
        self.prev_phase = torch.tensor(0.0)

    def make_argument(self, n_samples: int, freq: float, phase: float, device: torch.device) -> T:
        step = torch.tensor(2 * math.pi, dtype=torch.float32) * torch.tensor(freq, dtype=torch.float32) / self.sr
        k = torch.arange(1, n_samples + 1, dtype=torch.float64, device=device)
        csum = (k * float(step.double())).float()                      # = torch CPU cumsum of the constant fp32 step
        return csum + torch.tensor(phase, dtype=torch.float32, device=device)

    def forward(self, x: T, lfo_rate: T, lfo_depth: T, lfo_stereo_phase_offset: T) -> T:
        """x (2, 1, n) on the device; the three controls are 0-dim / 1-element tensors."""
        dev = x.device
        arg_l = self.make_argument(x.size(-1), float(lfo_rate), float(self.prev_phase), dev)
        self.prev_phase = (arg_l[-1] % (2 * math.pi)).cpu()
        arg_r = arg_l + torch.tensor(float(lfo_stereo_phase_offset), dtype=torch.float32, device=dev)
        lfo = (torch.cos(torch.stack([arg_l, arg_r], dim=0)) + 1.0) / 2.0
        lfo = (lfo * lfo_depth.to(dev)).unsqueeze(1)
        return self.model(x, lfo)


class EffectModelWrapper(nn.Module):
    """The control mapping of the Neutone wrapper: normalised knobs in [0, 1] -> rate 0.1-5 Hz, depth 0-1.5,
    stereo phase offset 0-2 pi; (2, n) in, (2, n) out."""

    def __init__(self, model: EffectModel) -> None:
        super().__init__()
        self.model = model

    def do_forward_pass(self, x: T, params: Dict[str, T]) -> T:
        lfo_rate = (params["lfo_rate"] * 4.9) + 0.1
        lfo_depth = params["lfo_depth"] * 1.5
        lfo_stereo_phase_offset = params["lfo_stereo_phase_offset"] * 2 * math.pi
        return self.model.forward(x.unsqueeze(1), lfo_rate, lfo_depth, lfo_stereo_phase_offset).squeeze(1)
