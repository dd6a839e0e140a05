"""Host mirror of mod_extraction/fx.py: same class, constructor and forward signature
(fx.py:25-44,121-130); the per-sample delay-line recurrence runs in the ``mx_flanger_fwd`` HIP
kernel (one wavefront per clip, delay line in LDS) instead of 88 200 python iterations.
"""
from typing import Dict, Optional, Tuple, Union

import torch
from torch import Tensor as T, nn

from . import _hip

Param = Union[float, T]


def delay_samples(ms: float, sr: float) -> int:
    """fx.py:40-41."""
    return int(((ms / 1000.0) * sr) + 0.5)


def apply_tremolo(x: T, mod_sig: T, mix: Param = 1.0) -> T:
    """fx.py:13-22 (a single fused elementwise expression; torch-on-HIP plumbing, not a kernel)."""
    assert x.ndim == 3 and x.size(0) == mod_sig.size(0) and x.size(-1) == mod_sig.size(-1)
    if mod_sig.ndim == 2:
        mod_sig = mod_sig.unsqueeze(1).expand(-1, x.size(1), -1)
    if isinstance(mix, T):
        assert mix.size(0) == x.size(0)
    assert 0.0 <= mix <= 1.0                    # fx.py:21 (a tensor mix must therefore hold one element)
    return ((1.0 - mix) * x) + (mix * mod_sig * x)


def _check_param(param: Param, bs: int, can_be_one: bool = True) -> None:
    """fx.py:46-70: all parameters >= 0; feedback < 1 strictly, the others <= 1."""
    if isinstance(param, T):
        assert param.shape == (bs,)
        lo, hi = float(param.min()), float(param.max())
    else:
        lo = hi = float(param)
    assert lo >= 0
    if can_be_one:
        assert hi <= 1.0
    else:
        assert hi < 1.0


def derive_clip_constants(bs: int, device: torch.device, max_min_delay_samples: int,
                          max_lfo_delay_samples: int, feedback: Param, min_delay_width: Param,
                          width: Param, depth: Param, mix: Param, check: bool = True) -> Dict[str, T]:
    """Per-clip fp32 constants of fx.py:98-99,114-117, rounded where the reference rounds them: a
    tensor parameter meets the integer sample count in fp32, a python float meets it in double."""
    if check:
        _check_param(feedback, bs, can_be_one=False)
        for p in (min_delay_width, width, depth, mix):
            _check_param(p, bs)

    def vec(p: Param) -> T:
        if isinstance(p, T):
            return p.to(device=device, dtype=torch.float32).contiguous()
        return torch.full((bs,), float(p), device=device, dtype=torch.float32)

    def times_int(p: Param, k: int) -> T:
        if isinstance(p, T):
            return (vec(p) * float(k)).contiguous()
        return torch.full((bs,), float(p) * k, device=device, dtype=torch.float32)

    mix_v = vec(mix)
    omm = (1.0 - mix_v) if isinstance(mix, T) else torch.full((bs,), 1.0 - float(mix), device=device,
                                                             dtype=torch.float32)
    return {"lfo_scale": times_int(width, max_lfo_delay_samples),
            "min_delay": times_int(min_delay_width, max_min_delay_samples),
            "feedback": vec(feedback), "depth": vec(depth), "mix": mix_v, "one_minus_mix": omm.contiguous()}


def _rows_view(t: T) -> Tuple[int, int]:
    """(data_ptr, row stride in floats) of a (B, N) view whose rows are contiguous."""
    if not t.is_cuda:
        raise _hip.HipLibraryError("mod_extraction_amd ops need tensors on a HIP device (no CPU fallback)")
    assert t.dtype == torch.float32 and t.ndim == 2 and t.stride(1) == 1
    return t.data_ptr(), t.stride(0)


def flanger_forward(x: T, mod_sig: T, consts: Dict[str, T], max_delay: T, max_delay_max: int,
                    rows: Optional[T] = None, out: Optional[T] = None, mod_up: Optional[T] = None,
                    dbg_prev: Optional[T] = None, dbg_frac: Optional[T] = None) -> T:
    """Launch mx_flanger_fwd.  x, out: (B,N) fp32 views with contiguous rows (any row stride, e.g. one
    channel of a (B,2,N) tensor); mod_sig (B,n_mod) fp32; max_delay (B,) int32."""
    B, N = x.shape
    y = out if out is not None else torch.empty_like(x)
    xp, xs = _rows_view(x)
    yp, ys = _rows_view(y)
    _hip.call("mx_flanger_fwd", xp, xs, _hip.ptr(mod_sig), mod_sig.size(-1),
              _hip.ptr(consts["lfo_scale"]), _hip.ptr(consts["min_delay"]), _hip.ptr(consts["feedback"]),
              _hip.ptr(consts["depth"]), _hip.ptr(consts["mix"]), _hip.ptr(consts["one_minus_mix"]),
              _hip.ptr(max_delay), int(max_delay_max), _hip.ptr(rows), 0 if rows is None else rows.numel(),
              B, N, yp, ys, _hip.ptr(mod_up), _hip.ptr(dbg_prev), _hip.ptr(dbg_frac), _hip.stream())
    return y


class MonoFlangerChorusModule(nn.Module):
    def __init__(self, batch_size: int, n_ch: int, n_samples: int, sr: float,
                 max_min_delay_ms: float, max_lfo_delay_ms: float) -> None:
        super().__init__()
        assert n_ch >= 1
        self.batch_size = batch_size
        self.n_ch = n_ch
        self.n_samples = n_samples
        self.sr = sr
        self.max_min_delay_ms = max_min_delay_ms
        self.max_lfo_delay_ms = max_lfo_delay_ms
        self.max_min_delay_samples = delay_samples(max_min_delay_ms, sr)
        self.max_lfo_delay_samples = delay_samples(max_lfo_delay_ms, sr)
        self.max_delay_samples = self.max_min_delay_samples + self.max_lfo_delay_samples
        # the reference registers delay_buf / out_buf buffers (fx.py:43-44); here the delay line
        # lives in LDS for the duration of the kernel and the output is the returned tensor.

    def check_param(self, param: Param, bs: int, out_n_dim: int = 2, can_be_one: bool = True) -> Param:
        """fx.py:46-70: range check of one effect parameter ((bs,) tensor or float in [0, 1], or [0, 1) when it may not be
        one) and its broadcast view; ``forward`` applies the same checks inside ``derive_clip_constants``."""
        if isinstance(param, T):
            assert param.shape == (bs,)
            assert param.min() >= 0
            assert param.max() <= 1.0 if can_be_one else param.max() < 1.0
            if out_n_dim not in (2, 3):
                raise ValueError
            return param.view((-1,) + (1,) * (out_n_dim - 1))
        assert param >= 0
        assert param <= 1.0 if can_be_one else param < 1.0
        return param

    def apply_effect(self, x: T, mod_sig: T, feedback: Param, min_delay_width: Param, width: Param, depth: Param,
                     mix: Param) -> T:
        """fx.py:72-119 (the per-sample loop) = one kernel launch here; ``forward`` is this under no_grad, as in the
        reference."""
        return self.forward(x, mod_sig, feedback, min_delay_width, width, depth, mix)

    def forward(self, x: T, mod_sig: T, feedback: Param = 0.0, min_delay_width: Param = 1.0,
                width: Param = 1.0, depth: Param = 1.0, mix: Param = 1.0) -> T:
        assert x.ndim == 3
        bs, n_ch, n = x.shape
        assert n_ch == self.n_ch
        assert mod_sig.size(0) == bs
        if mod_sig.ndim == 3:
            assert mod_sig.size(1) in (1, n_ch)
        with torch.no_grad():
            consts = derive_clip_constants(bs, x.device, self.max_min_delay_samples,
                                           self.max_lfo_delay_samples, feedback, min_delay_width,
                                           width, depth, mix)
            # n_ch > 1 (fx.py:81-85,104-115): every channel owns a delay line = one kernel row per (clip, channel); a clip's
            # channels share its parameters; mod_sig (bs, n) / (bs, 1, n) is shared by the channels
            rows = bs * n_ch
            if n_ch > 1:
                consts = {k: v.repeat_interleave(n_ch) for k, v in consts.items()}
                mod_sig = mod_sig.view(bs, 1, -1).expand(-1, n_ch, -1) if mod_sig.ndim == 2 or mod_sig.size(1) == 1 else mod_sig
            md = torch.full((rows,), self.max_delay_samples, device=x.device, dtype=torch.int32)
            xc = x.reshape(rows, n).contiguous().float()
            mc = mod_sig.reshape(rows, -1).contiguous().float()
            y = flanger_forward(xc, mc, consts, md, self.max_delay_samples)
        return y.view(bs, n_ch, n)


def phaser_forward(src: T, params: Dict[str, T], lead: Optional[T], sr: float, n_samples: int,
                   rows: Optional[T] = None, out: Optional[T] = None, dry_out: Optional[T] = None,
                   exact_order: bool = False) -> T:
    """Launch mx_phaser_fwd (pedalboard.Phaser semantics, datasets.py:455-482).
    src (B, >= lead+n_samples) source audio rows; params: rate_hz, depth, centre_frequency_hz,
    feedback, mix -- each (B,) fp32 on the device; lead (B,) int32 warm-up samples or None;
    out / dry_out: (B, n_samples) views with contiguous rows.  exact_order=True runs every sample in JUCE's
    operation order on one wavefront per clip (the bit reference, ~20x slower); the default is the time-parallel scan."""
    B = src.size(0)
    y = out if out is not None else torch.empty((B, n_samples), device=src.device, dtype=torch.float32)
    sp, ss = _rows_view(src)
    yp, ys = _rows_view(y)
    dp = None
    if dry_out is not None:
        dp, ds = _rows_view(dry_out)
        assert ds == ys
    # cut-off workspace of the scan kernel: one float per 4-sample group of the longest possible render (lead + n_samples
    # never exceeds a source row)
    n_items = B if rows is None else rows.numel()
    ws_stride = (src.size(-1) + 3) // 4
    ws = None if exact_order else torch.empty((n_items, ws_stride), device=src.device, dtype=torch.float32)
    _hip.call("mx_phaser_fwd", sp, ss, _hip.ptr(params["rate_hz"]), _hip.ptr(params["depth"]),
              _hip.ptr(params["centre_frequency_hz"]), _hip.ptr(params["feedback"]), _hip.ptr(params["mix"]),
              _hip.ptr(lead), _hip.ptr(rows), 0 if rows is None else rows.numel(), B, n_samples, float(sr),
              1 if exact_order else 0, yp, ys, dp, _hip.ptr(ws), ws_stride, _hip.stream())
    return y
