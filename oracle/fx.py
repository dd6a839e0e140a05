"""Oracle restatement of mod_extraction/fx.py (TEST INFRASTRUCTURE ONLY).

The per-sample recurrence lives in oracle/csrc/oracle_ref.c (``orc_flanger``), compiled with
``-ffp-contract=off``; this file restates the parameter handling of fx.py:46-90 and the module
surface of fx.py:25-44,121-130.  Pinned BIT-FOR-BIT against the reference's own
``fx.MonoFlangerChorusModule`` (importable) by tests/golden/make_golden.py -> flanger.npz.
"""
from typing import Dict, Optional, Union

import numpy as np
import torch

from . import _cref

F32 = np.float32
Param = Union[float, torch.Tensor]


def delay_samples(ms: float, sr: float) -> int:
    """fx.py:40-41."""
    return int(((ms / 1000.0) * sr) + 0.5)


def _check(param: Param, bs: int, can_be_one: bool = True) -> None:
    """Range rules of fx.py:46-70 (all >= 0; feedback strictly < 1, the rest <= 1)."""
    if isinstance(param, torch.Tensor):
        assert param.shape == (bs,)
        lo, hi = float(param.min()), float(param.max())
    else:
        lo = hi = param
    assert lo >= 0
    assert hi <= 1.0 if can_be_one else hi < 1.0


def _as_f32(param: Param, bs: int) -> np.ndarray:
    if isinstance(param, torch.Tensor):
        return param.detach().cpu().numpy().astype(F32)
    return np.full((bs,), F32(param), dtype=F32)       # python float -> fp32 scalar when it meets a tensor


def derive_params(bs: int, max_min_delay_samples: int, max_lfo_delay_samples: int,
                  feedback: Param, min_delay_width: Param, width: Param, depth: Param,
                  mix: Param) -> Dict[str, np.ndarray]:
    """Per-clip fp32 constants, rounded exactly where fx.py:98-99,114-117 rounds them.

    A python-float parameter is combined with the integer sample count in *double* and only then
    rounded to fp32 (python arithmetic precedes the tensor op); a tensor parameter is multiplied in
    fp32.  Both paths are kept because both occur at the reference's call sites.
    """
    _check(feedback, bs, can_be_one=False)
    for p in (min_delay_width, width, depth, mix):
        _check(p, bs)

    def times_int(p: Param, k: int) -> np.ndarray:
        if isinstance(p, torch.Tensor):
            return (_as_f32(p, bs) * F32(k)).astype(F32)
        return np.full((bs,), F32(float(p) * k), dtype=F32)

    if isinstance(mix, torch.Tensor):
        one_minus_mix = (F32(1.0) - _as_f32(mix, bs)).astype(F32)
    else:
        one_minus_mix = np.full((bs,), F32(1.0 - float(mix)), dtype=F32)
    return {
        "lfo_scale": times_int(width, max_lfo_delay_samples),
        "min_delay": times_int(min_delay_width, max_min_delay_samples),
        "feedback": _as_f32(feedback, bs),
        "depth": _as_f32(depth, bs),
        "mix": _as_f32(mix, bs),
        "one_minus_mix": one_minus_mix,
    }


def flanger_np(x: np.ndarray, mod: np.ndarray, p: Dict[str, np.ndarray], M: int,
               want_indices: bool = False):
    """x, mod: (B, N) fp32.  Returns y (B, N) [and prev_idx int64, frac fp32 if asked]."""
    x = np.ascontiguousarray(x, dtype=F32)
    mod = np.ascontiguousarray(mod, dtype=F32)
    B, N = x.shape
    y = np.empty_like(x)
    prev = np.empty((B, N), dtype=np.int64) if want_indices else None
    frac = np.empty((B, N), dtype=F32) if want_indices else None
    args = [np.ascontiguousarray(p[k], dtype=F32) for k in
            ("lfo_scale", "min_delay", "feedback", "depth", "mix", "one_minus_mix")]
    _cref.lib().orc_flanger(_cref.fptr(x), _cref.fptr(mod), *[_cref.fptr(a) for a in args],
                            B, N, M, _cref.fptr(y), _cref.iptr(prev), _cref.fptr(frac))
    return (y, prev, frac) if want_indices else y


class MonoFlangerChorusModule(torch.nn.Module):
    """Same constructor / forward surface as fx.py:25-130.  n_ch > 1 (fx.py:81-85,104-115): every channel owns a delay
    line, a clip's channels share its parameters, mod_sig is (bs, n) -- shared -- or (bs, n_ch, n)."""

    def __init__(self, batch_size: int, n_ch: int, n_samples: int, sr: float,
                 max_min_delay_ms: float, max_lfo_delay_ms: float) -> None:
        super().__init__()
        assert n_ch >= 1
        self.batch_size, self.n_ch, self.n_samples, self.sr = batch_size, n_ch, n_samples, sr
        self.max_min_delay_ms, self.max_lfo_delay_ms = max_min_delay_ms, max_lfo_delay_ms
        self.max_min_delay_samples = delay_samples(max_min_delay_ms, sr)
        self.max_lfo_delay_samples = delay_samples(max_lfo_delay_ms, sr)
        self.max_delay_samples = self.max_min_delay_samples + self.max_lfo_delay_samples

    @torch.no_grad()
    def forward(self, x: torch.Tensor, mod_sig: torch.Tensor, feedback: Param = 0.0,
                min_delay_width: Param = 1.0, width: Param = 1.0, depth: Param = 1.0,
                mix: Param = 1.0) -> torch.Tensor:
        assert x.ndim == 3 and x.size(1) == self.n_ch
        bs, c, n = x.shape
        assert mod_sig.size(0) == bs and mod_sig.size(-1) == n
        p = derive_params(bs, self.max_min_delay_samples, self.max_lfo_delay_samples,
                          feedback, min_delay_width, width, depth, mix)
        if mod_sig.ndim == 2:
            mod_sig = mod_sig.unsqueeze(1).expand(-1, c, -1)                        # fx.py:84-85
        p = {k: np.repeat(v, c) for k, v in p.items()}                              # a clip's channels share its parameters
        y = flanger_np(np.ascontiguousarray(x.reshape(bs * c, n).numpy()), np.ascontiguousarray(mod_sig.reshape(bs * c, n).numpy()), p,
                       self.max_delay_samples)
        return torch.from_numpy(y).view(bs, c, n)


def apply_tremolo(x: torch.Tensor, mod_sig: torch.Tensor, mix: Param = 1.0) -> torch.Tensor:
    """fx.py:13-22."""
    assert x.ndim == 3 and x.size(0) == mod_sig.size(0) and x.size(-1) == mod_sig.size(-1)
    if mod_sig.ndim == 2:
        mod_sig = mod_sig.unsqueeze(1).expand(-1, x.size(1), -1)
    return ((1.0 - mix) * x) + (mix * mod_sig * x)


def phaser_np(x: np.ndarray, rate, depth, centre, feedback, mix, sr: float, want_lfo: bool = False):
    """pedalboard.Phaser restatement (PARITY UNPINNED): x (B, N) fp32, params (B,) fp32."""
    x = np.ascontiguousarray(x, dtype=F32)
    B, N = x.shape
    y = np.empty_like(x)
    lfo: Optional[np.ndarray] = np.empty((B, (N + 3) // 4), dtype=F32) if want_lfo else None
    ps = [np.ascontiguousarray(np.broadcast_to(np.asarray(v, dtype=F32), (B,))) for v in
          (rate, depth, centre, feedback, mix)]
    _cref.lib().orc_phaser(_cref.fptr(x), *[_cref.fptr(a) for a in ps], B, N, float(sr),
                           _cref.fptr(y), _cref.fptr(lfo))
    return (y, lfo) if want_lfo else y


def flanger_torch_loop(x: torch.Tensor, mod_sig: torch.Tensor, M_min: int, M_lfo: int, feedback: torch.Tensor,
                       min_delay_width: torch.Tensor, width: torch.Tensor, depth: torch.Tensor,
                       mix: torch.Tensor) -> torch.Tensor:
    """The reference's OWN execution shape of fx.py:92-119: index tensors for the whole clip up front, then one
    python iteration per sample, each a handful of tiny batched torch ops on the (B, 1, M) delay line.  Used only
    (a) by bench.py's ``cpu_baseline.reference_shaped`` leg to time what the reference pays per sample, and
    (b) by tests/test_oracle_golden.py to cross-check the C restatement (bit-identical).
    x (B, 1, N), mod_sig (B, N), parameters (B,) fp32."""
    B, C, N = x.shape
    M = M_min + M_lfo
    ring = torch.zeros(B, C, M)
    acc = torch.zeros(B, C, N)
    w_pos = (torch.arange(N) % M).view(1, 1, N).expand(B, C, N)
    base = min_delay_width.view(B, 1, 1) * M_min
    delay = (M_lfo * width.view(B, 1, 1) * mod_sig.unsqueeze(1).expand(B, C, N)) + base
    r_pos = (w_pos - delay + M) % M
    lo = torch.floor(r_pos)
    frac = r_pos - lo
    lo = lo.to(torch.long)
    hi = (lo + 1) % M
    fb, dp = feedback.view(B, 1), depth.view(B, 1)
    for n in range(N):
        dry_n = x[:, :, n]
        a = torch.gather(ring, -1, lo[:, :, n].unsqueeze(-1)).squeeze(-1)
        b = torch.gather(ring, -1, hi[:, :, n].unsqueeze(-1)).squeeze(-1)
        f = frac[:, :, n]
        tap = (f * b) + ((1.0 - f) * a)
        ring[:, :, w_pos[0, 0, n]] = dry_n + (fb * tap)
        acc[:, :, n] = dry_n + (dp * tap)
    m = mix.view(B, 1, 1)
    return torch.clip(((1.0 - m) * x) + (m * acc), -1.0, 1.0)
