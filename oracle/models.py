"""Oracle restatement of mod_extraction/models.py (TEST INFRASTRUCTURE ONLY), torch on the CPU.

* ``mel_frontend`` restates ``torchaudio.transforms.MelSpectrogram`` (torchaudio==0.13.1, absent
  from /root/reference and from this image) as used at models.py:170-181,199-208 -- from its
  published algorithm: ``torch.stft`` (hann periodic window, centre, reflect pad) -> ``abs().pow(2)``
  -> HTK triangular filter bank without normalisation -> optional SpecAugment masks -> clip -> log.
  The STFT is torch's own (pinned); the filter-bank construction and masking are PARITY UNPINNED.
* ``Spectral2DCNN`` restates models.py:128-215 with the same ``torch.nn`` modules in the same
  ``nn.Sequential`` positions, so state-dict keys match the reference
  (``cnn.{1,5,..}.weight|bias``, ``cnn.{3,7,..}.weight``, ``output.weight|bias`` and the buffers
  ``spectrogram.spectrogram.window``, ``spectrogram.mel_scale.fb``).  Pinned against the
  reference's own ``models.Spectral2DCNN.cnn``/``output`` stack (imported with a name-only
  torchaudio stub) by tests/golden/make_golden_nn.py.
* ``LSTMEffectModel`` restates models.py:292-339; pinned with the 7 shipped LSTM-64 weight files.
"""
import math
from typing import List, Optional, Sequence, Tuple

import torch
from torch import Tensor as T, nn


# ---- torchaudio.functional.melscale_fbanks(norm=None, mel_scale="htk") -----------------------
def htk_mel_filterbank(n_freqs: int, f_min: float, f_max: float, n_mels: int, sample_rate: int) -> T:
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_lo = 2595.0 * math.log10(1.0 + (f_min / 700.0))
    m_hi = 2595.0 * math.log10(1.0 + (f_max / 700.0))
    m_pts = torch.linspace(m_lo, m_hi, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)          # (n_freqs, n_mels + 2)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.max(torch.zeros(1), torch.min(down, up))         # (n_freqs, n_mels)


def specaugment_bounds(size: int, mask_param: int) -> Tuple[int, int]:
    """torchaudio.functional.mask_along_axis (one mask for the whole batch): two torch.rand(1)
    draws, ``value`` then ``min_value``; masked range is [int(min_value), int(min_value)+int(value))."""
    value = torch.rand(1) * mask_param
    min_value = torch.rand(1) * (size - value)
    start = int(min_value.long())
    return start, start + int(value.long())


class _Holder(nn.Module):
    """Namespace module so buffers get the torchaudio key names."""


class MelFrontEnd(nn.Module):
    def __init__(self, sr: int, n_fft: int, hop_len: int, n_mels: int) -> None:
        super().__init__()
        self.n_fft, self.hop_len, self.n_mels = n_fft, hop_len, n_mels
        self.spectrogram = _Holder()
        self.spectrogram.register_buffer("window", torch.hann_window(n_fft))
        self.mel_scale = _Holder()
        self.mel_scale.register_buffer("fb", htk_mel_filterbank(n_fft // 2 + 1, 0.0, float(sr // 2), n_mels, sr))

    def forward(self, x: T) -> T:
        shape = x.shape
        spec = torch.stft(x.reshape(-1, shape[-1]), self.n_fft, self.hop_len, self.n_fft,
                          self.spectrogram.window, center=True, pad_mode="reflect", normalized=False,
                          onesided=True, return_complex=True)
        power = spec.abs().pow(2.0)
        power = power.reshape(shape[:-1] + power.shape[-2:])       # (B, C, n_freqs, frames)
        return torch.matmul(power.transpose(-1, -2), self.mel_scale.fb).transpose(-1, -2)


class Spectral2DCNN(nn.Module):
    def __init__(self, in_ch: int = 1, n_samples: int = 88200, sr: float = 44100, n_fft: int = 1024,
                 hop_len: int = 256, n_mels: int = 256, kernel_size: Tuple[int, int] = (5, 13),
                 out_channels: Optional[List[int]] = None, bin_dilations: Optional[List[int]] = None,
                 temp_dilations: Optional[List[int]] = None, pool_size: Tuple[int, int] = (3, 1),
                 latent_dim: int = 1, freq_mask_amount: float = 0.0, time_mask_amount: float = 0.0,
                 use_ln: bool = True, eps: float = 1e-7) -> None:
        super().__init__()
        out_channels = [64] * 5 if out_channels is None else list(out_channels)
        bin_dilations = [1] * len(out_channels) if bin_dilations is None else list(bin_dilations)
        temp_dilations = [2 ** i for i in range(len(out_channels))] if temp_dilations is None else list(temp_dilations)
        assert pool_size[1] == 1 and len(out_channels) == len(bin_dilations) == len(temp_dilations)
        self.eps, self.latent_dim = eps, latent_dim
        self.n_frames = n_samples // hop_len + 1
        self.freq_mask_param = int(freq_mask_amount * n_mels)
        self.time_mask_param = int(time_mask_amount * self.n_frames)
        self.freq_mask_amount, self.time_mask_amount = freq_mask_amount, time_mask_amount
        self.spectrogram = MelFrontEnd(int(sr), n_fft, hop_len, n_mels)
        stack, bins, c_in = [], n_mels, in_ch
        for c_out, bd, td in zip(out_channels, bin_dilations, temp_dilations):
            if use_ln:
                stack.append(nn.LayerNorm([bins, self.n_frames], elementwise_affine=False))
            stack += [nn.Conv2d(c_in, c_out, tuple(kernel_size), stride=(1, 1), dilation=(bd, td), padding="same"),
                      nn.MaxPool2d(kernel_size=tuple(pool_size)), nn.PReLU(num_parameters=c_out)]
            c_in, bins = c_out, bins // pool_size[0]
        self.cnn = nn.Sequential(*stack)
        self.output = nn.Conv1d(out_channels[-1], latent_dim, kernel_size=(1,))

    def log_mel(self, x: T, masks: Optional[Sequence[int]] = None) -> T:
        """models.py:199-208.  ``masks`` = (f0, f1, t0, t1) injects the SpecAugment ranges (tests);
        None in training mode draws them like torchaudio does."""
        m = self.spectrogram(x)
        if masks is None and self.training:
            f0 = f1 = t0 = t1 = 0
            if self.freq_mask_amount > 0:
                f0, f1 = specaugment_bounds(m.size(-2), self.freq_mask_param)
            if self.time_mask_amount > 0:
                t0, t1 = specaugment_bounds(m.size(-1), self.time_mask_param)
            masks = (f0, f1, t0, t1)
        if masks is not None:
            f0, f1, t0, t1 = masks
            m = m.clone()
            m[..., f0:f1, :] = 0.0
            m[..., :, t0:t1] = 0.0
        return torch.log(torch.clip(m, min=self.eps))

    def forward(self, x: T, masks: Optional[Sequence[int]] = None) -> Tuple[T, T]:
        assert x.ndim == 3
        latent = torch.mean(self.cnn(self.log_mel(x, masks)), dim=-2)      # models.py:209-211
        return torch.sigmoid(self.output(latent)), latent                 # models.py:213-215


class LSTMEffectModel(nn.Module):
    """models.py:292-339 (HiddenStateModel + LSTMEffectModel)."""

    def __init__(self, in_ch: int = 1, out_ch: int = 1, n_hidden: int = 64, latent_dim: int = 1) -> None:
        super().__init__()
        self.in_ch, self.out_ch, self.n_hidden, self.latent_dim = in_ch, out_ch, n_hidden, latent_dim
        self.lstm = nn.LSTM(in_ch + latent_dim, n_hidden, batch_first=True)
        self.fc = nn.Linear(n_hidden, out_ch)
        self.hidden: Optional[Tuple[T, T]] = None

    def clear_hidden(self) -> None:
        self.hidden = None

    def detach_hidden(self) -> None:
        if self.hidden is not None:
            self.hidden = tuple(h.detach().clone() for h in self.hidden)

    def forward(self, x: T, latent: T) -> T:
        assert x.ndim == 3 and latent.shape == (x.size(0), self.latent_dim, x.size(-1))
        seq = torch.cat([latent, x], dim=1).swapaxes(1, 2)               # LFO first, audio second
        out, self.hidden = self.lstm(seq, self.hidden)
        return torch.tanh(self.fc(out).swapaxes(1, 2) + x)


def forward_routed(ref: "Spectral2DCNN", x: T, masks, tap: dict, W: int, tie_tol: float = 2e-6, kink_stats: dict = None):
    """Oracle forward that takes the device's decisions at the two non-differentiable points of a block: the
    MaxPool2d((2,1)) argmax (``tap["amax<l>"]``) and the PReLU branch (sign of ``tap["p<l>"]``).  Wherever a
    decision differs from torch's own, the oracle's values must sit on the kink to fp32 rounding (the two pooled
    rows equal, or the pre-activation ~0; ``tie_tol`` relative to the tensor's max): either side is a valid
    sub-gradient there.  Sharing the decisions lets every downstream gradient be compared at fp32 tolerance.
    Returns (sigmoid output, latent, number of shared kink decisions).  ``kink_stats`` (optional dict) accumulates how
    wide the shared decisions were: ``n`` of them, ``n_wide`` with |delta| above the one-step rule (2e-6 of the tensor's
    max), ``max_rel`` the widest -- a test that allows a looser ``tie_tol`` prints these."""
    h = ref.log_mel(x, masks)
    n_kinks = 0

    def note(gap: T, scale: float) -> None:
        if kink_stats is not None and gap.numel():
            rel = gap / max(scale, 1e-30)
            kink_stats["n"] = kink_stats.get("n", 0) + int(rel.numel())
            kink_stats["n_wide"] = kink_stats.get("n_wide", 0) + int((rel > 2e-6).sum())
            kink_stats["max_rel"] = max(kink_stats.get("max_rel", 0.0), float(rel.max()))
    for i, m in enumerate(ref.cnn):
        blk = i // 4
        if isinstance(m, nn.MaxPool2d):
            top, bot = h[:, :, 0::2], h[:, :, 1::2]
            pick = tap[f"amax{blk}"].cpu()[..., :W].bool()
            diff = pick != (bot > top)
            if diff.any():
                assert float((top - bot).abs()[diff].max()) <= tie_tol * float(h.detach().abs().max()), \
                    "argmax differs away from a tie"
                note((top - bot).detach().abs()[diff], float(h.detach().abs().max()))
                n_kinks += int(diff.sum())
            h = torch.where(pick, bot, top)
        elif isinstance(m, nn.PReLU):
            pos = tap[f"p{blk}"].cpu()[..., :W] > 0
            diff = pos != (h > 0)
            if diff.any():
                assert float(h.detach().abs()[diff].max()) <= tie_tol * float(h.detach().abs().max()), \
                    "PReLU branch differs away from zero"
                note(h.detach().abs()[diff], float(h.detach().abs().max()))
                n_kinks += int(diff.sum())
            h = torch.where(pos, h, m.weight.view(1, -1, 1, 1) * h)
        else:
            h = m(h)
    latent = h.mean(dim=-2)
    return torch.sigmoid(ref.output(latent)), latent, n_kinks


# ---- f4: the TCN extractors (models.py:72-125, 218-289) --------------------------------------------------------
class _PowerSpectrogram(nn.Module):
    """torchaudio.transforms.Spectrogram(n_fft, hop_length=hop, normalized=False): power 2, periodic hann, centre +
    reflect padding (third-party, restated through torch.stft; its ``window`` buffer keeps the state-dict key)."""

    def __init__(self, n_fft: int, hop: int) -> None:
        super().__init__()
        self.n_fft, self.hop = n_fft, hop
        self.register_buffer("window", torch.hann_window(n_fft))

    def forward(self, x: T) -> T:
        s = torch.stft(x.reshape(-1, x.size(-1)), self.n_fft, self.hop, self.n_fft, self.window, center=True,
                       pad_mode="reflect", normalized=False, return_complex=True)
        return (s.real ** 2 + s.imag ** 2).view(x.shape[:-1] + s.shape[-2:])


class SpectralTCN(nn.Module):
    def __init__(self, n_samples: int = 88200, n_fft: int = 1024, hop_len: int = 256, kernel_size: int = 13,
                 out_channels=None, dilations=None, latent_dim: int = 1, use_ln: bool = True, use_res: bool = True,
                 eps: float = 1e-7) -> None:
        super().__init__()
        from .tcn import TCN
        out_channels = out_channels or [96] * 5
        dilations = dilations or [2 ** i for i in range(len(out_channels))]
        self.eps = eps
        self.spectrogram = _PowerSpectrogram(n_fft, hop_len)
        n_frames = n_samples // hop_len + 1
        self.tcn = TCN(out_channels, dilations, n_fft // 2 + 1, kernel_size, None, use_ln, [n_frames] * len(out_channels),
                       True, use_res)
        self.output = nn.Conv1d(out_channels[-1], latent_dim, kernel_size=(1,))

    def forward(self, x: T) -> T:
        x = torch.log(torch.clip(self.spectrogram(x).squeeze(1), min=self.eps))      # models.py:118-121
        return torch.sigmoid(self.output(self.tcn(x)))


class SpectralDSTCN(nn.Module):
    def __init__(self, n_samples: int = 88200, n_fft: int = 1024, hop_len: int = 256, kernel_size: int = 13,
                 out_channels=None, dilations=None, strides=None, n_fc_units: int = 48, latent_dim: int = 2,
                 use_ln: bool = True, use_res: bool = True, eps: float = 1e-7) -> None:
        super().__init__()
        import math
        from .tcn import TCN
        out_channels = out_channels or [96] * 5
        dilations = dilations or [2 ** i for i in range(len(out_channels))]
        strides = strides or [2] * len(out_channels)
        self.eps = eps
        self.spectrogram = _PowerSpectrogram(n_fft, hop_len)
        dims, cur = [n_samples // hop_len + 1], n_samples // hop_len + 1
        for s in strides[:-1]:                                                       # models.py:255-260
            cur = math.ceil(cur / s)
            dims.append(cur)
        self.tcn = TCN(out_channels, dilations, n_fft // 2 + 1, kernel_size, strides, use_ln, dims, True, use_res)
        self.fc = nn.Linear(out_channels[-1], n_fc_units)
        self.fc_act = nn.PReLU(n_fc_units)
        self.output = nn.Linear(n_fc_units, latent_dim)

    def forward(self, x: T) -> T:
        x = torch.log(torch.clip(self.spectrogram(x).squeeze(1), min=self.eps))
        x = self.tcn(x).mean(dim=-1)                                                 # models.py:282-283
        return torch.sigmoid(self.output(self.fc_act(self.fc(x))))
