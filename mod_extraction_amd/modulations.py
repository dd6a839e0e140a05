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


# ---- K9: corner bookkeeping (modulations.py:219-363), bit-exact on the device ------------------
def _rows2d(x: T):
    assert x.ndim == 2
    return x.contiguous().float()


def smoothen(x: T, smooth_n_frames: int) -> T:
    """modulations.py:359-363."""
    if smooth_n_frames <= 1:
        return x
    lead = x.shape[:-1]
    xc = x.reshape(-1, x.size(-1)).contiguous().float()
    n_out = xc.size(-1) - smooth_n_frames + 1
    out = torch.empty((xc.size(0), n_out), device=xc.device, dtype=torch.float32)
    _hip.call("mx_smoothen", _hip.ptr(xc), xc.size(0), xc.size(-1), smooth_n_frames, _hip.ptr(out), _hip.stream())
    return out.view(lead + (n_out,))


def find_corners(mod_sig: T) -> (T, T):
    """modulations.py:219-238: float 0/1 maps of top and bottom corners."""
    m = _rows2d(mod_sig)
    top, bot = torch.empty_like(m), torch.empty_like(m)
    _hip.call("mx_find_corners", _hip.ptr(m), m.size(0), m.size(1), _hip.ptr(top), _hip.ptr(bot), _hip.stream())
    return top, bot


def stretch_corners(mod_sig: T, max_n_corners: int = 10, smooth_n_frames: int = 32) -> T:
    """modulations.py:294-307."""
    m = _rows2d(smoothen(mod_sig, smooth_n_frames))
    out = torch.empty_like(m)
    _hip.call("mx_stretch_corners", _hip.ptr(m), m.size(0), m.size(1), int(max_n_corners), _hip.ptr(out),
              _hip.stream())
    return out


class _SmoothenFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: T, k: int):
        ctx.k = k
        return smoothen(x.detach(), k)

    @staticmethod
    def backward(ctx, dy: T):
        return smoothen_bwd(dy.contiguous(), ctx.k), None


def smoothen_with_grad(x: T, smooth_n_frames: int) -> T:
    """``smoothen`` as an autograd node (forward ``mx_smoothen``, backward its transpose): lightning.py:117-119 while training."""
    return x if smooth_n_frames <= 1 else _SmoothenFn.apply(x, smooth_n_frames)


def smoothen_bwd(dy: T, smooth_n_frames: int) -> T:
    """Transpose of ``smoothen``: the same moving average over the zero-padded gradient."""
    if smooth_n_frames <= 1:
        return dy
    k = smooth_n_frames
    return smoothen(torch.nn.functional.pad(dy, (k - 1, k - 1)), k)


def stretch_corners_bwd(mod_sig: T, d_out: T, max_n_corners: int = 10, smooth_n_frames: int = 32) -> T:
    """Gradient of ``stretch_corners(mod_sig, ...)`` w.r.t. ``mod_sig`` given the gradient w.r.t. its result (the reference's
    _stretch_corners is differentiable torch code, modulations.py:260-291; corner positions carry no gradient)."""
    m = _rows2d(smoothen(mod_sig, smooth_n_frames))
    g = _rows2d(d_out)
    assert g.shape == m.shape
    dm = torch.empty_like(m)
    _hip.call("mx_stretch_corners_bwd", _hip.ptr(m), _hip.ptr(g), m.size(0), m.size(1), int(max_n_corners), _hip.ptr(dm),
              _hip.stream())
    return smoothen_bwd(dm, smooth_n_frames)


def valid_mod_sig_mask(mod_sig: T, min_top_corners: int = 1, max_top_corners: int = 6,
                       min_bottom_corners: int = 1, max_bottom_corners: int = 6,
                       min_fraction_between_corners: float = 0.10) -> T:
    """check_mod_sig (modulations.py:311-343) for every row: (B,) int32 0/1 on the device."""
    m = _rows2d(mod_sig)
    valid = torch.empty(m.size(0), device=m.device, dtype=torch.int32)
    _hip.call("mx_check_mod_sig", _hip.ptr(m), m.size(0), m.size(1), min_top_corners, max_top_corners,
              min_bottom_corners, max_bottom_corners, int(min_fraction_between_corners * m.size(1)),
              _hip.ptr(valid), _hip.stream())
    return valid


def mod_sig_to_corners(mod_sig: T, n_frames: int) -> (T, T):
    """modulations.py:212-216: corners of the LFOs resampled to ``n_frames`` points."""
    from . import util
    assert mod_sig.ndim == 2
    return find_corners(util.linear_interpolate_last_dim(mod_sig, n_frames))


def check_mod_sig(mod_sig: T, top_corners: T, bottom_corners: T, min_top_corners: int = 1, max_top_corners: int = 6,
                  min_bottom_corners: int = 1, max_bottom_corners: int = 6,
                  min_fraction_between_corners: float = 0.10) -> bool:
    """modulations.py:311-343 for ONE LFO and the corner maps the caller holds (the batched form on the device is
    ``valid_mod_sig_mask``): corner counts within bounds and neighbouring corners of a kind at least
    ``int(min_fraction_between_corners * n)`` frames apart."""
    assert mod_sig.ndim == 1 and mod_sig.shape == top_corners.shape == bottom_corners.shape
    min_gap = int(min_fraction_between_corners * mod_sig.size(0))
    for corners, lo, hi in ((top_corners, min_top_corners, max_top_corners),
                            (bottom_corners, min_bottom_corners, max_bottom_corners)):
        if not lo <= int(corners.sum()) <= hi:
            return False
    for corners in (top_corners, bottom_corners):
        at = torch.nonzero(corners == 1).view(-1)
        if at.numel() > 1 and int((at[1:] - at[:-1]).min()) < min_gap:
            return False
    return True


def corners_to_mod_sig(top_corners: T, bottom_corners: T) -> T:
    """modulations.py:241-257: the triangle LFO through the corners of ONE row -- 1 at the top corners, 0 at the bottom
    ones, straight lines in between, 0 outside the first / last corner and when either kind is absent."""
    assert top_corners.ndim == 1 and top_corners.shape == bottom_corners.shape
    out = torch.zeros(top_corners.shape, dtype=torch.float32, device=top_corners.device)
    if float(top_corners.max()) == 0 or float(bottom_corners.max()) == 0:
        return out
    knots = sorted([(int(i), 1.0) for i in torch.nonzero(top_corners == 1).view(-1)]
                   + [(int(i), 0.0) for i in torch.nonzero(bottom_corners == 1).view(-1)])
    for (l, lv), (r, rv) in zip(knots[:-1], knots[1:]):
        out[l:r + 1] = torch.linspace(lv, rv, r - l + 1, device=out.device)
    return out


def find_valid_mod_sig_indices(mod_sig: T) -> List[int]:
    """modulations.py:346-356 (host list, like the reference; one small D2H copy)."""
    return torch.nonzero(valid_mod_sig_mask(mod_sig)).view(-1).tolist()


def make_rand_mod_signal(batch_size: int, n_samples: int, sr: float, freq_min: float, freq_max: float,
                         shapes_gt: Optional[Sequence[str]] = None, shapes: Optional[List[str]] = None,
                         phase_gt: Optional[T] = None, phase_error: float = 0.5,
                         freq_gt: Optional[T] = None, freq_error: float = 0.25, device=None) -> T:
    """modulations.py:60-101: the per-item host RNG draws are kept in the reference's order; the
    batch of LFOs is then synthesised by one ``mx_lfo_synth`` launch."""
    from . import util
    if shapes is None:
        shapes = ["cos", "tri", "rect_cos", "inv_rect_cos", "saw", "rsaw"]
    freqs, phases, shape_ids = [], [], []
    # ground-truth rows are perturbed with the reference's own fp32 tensor arithmetic (0-dim CPU tensors, in-place
    # add / multiply, then the wrap / clip), on private copies: the reference mutates the caller's fx_params here
    phase_gt = None if phase_gt is None else phase_gt.detach().to("cpu", torch.float32).clone()
    freq_gt = None if freq_gt is None else freq_gt.detach().to("cpu", torch.float32).clone()
    for idx in range(batch_size):
        if phase_gt is not None:
            assert phase_gt.size(0) == batch_size
            phase = phase_gt[idx]
            if phase_error > 0:
                phase += util.sample_uniform(-1.0, 1.0) * math.pi * phase_error
                phase = (phase + (2 * math.pi)) % (2 * math.pi)
        else:
            phase = util.sample_uniform(0.0, 2 * math.pi)
        if freq_gt is not None:
            assert freq_gt.size(0) == batch_size
            freq = freq_gt[idx]
            if freq_error > 0:
                freq *= util.sample_uniform(1.0 - freq_error, 1.0 + freq_error)
                freq = torch.clip(freq, freq_min, freq_max)
        else:
            freq = util.sample_uniform(freq_min, freq_max)
        shape = shapes_gt[idx] if shapes_gt is not None else util.choice(shapes)
        freqs.append(float(freq)); phases.append(float(phase)); shape_ids.append(SHAPE_IDS[shape])
    dev = _device(device)
    return make_mod_signals(n_samples, sr, torch.tensor(freqs, dtype=torch.float32, device=dev),
                            torch.tensor(phases, dtype=torch.float32, device=dev),
                            torch.tensor(shape_ids, dtype=torch.int32, device=dev))


# ---- evaluation LFO variants (modulations.py:104-210) ------------------------------------------
# Per-item data-generation helpers of the eval configs (eval_lfo_quasi / combined / distorted): host
# control flow and host RNG draws in the reference's order; every array operation (LFO synthesis,
# corner detection, resampling) runs in the device kernels above.
def _time_stretch_section(section: T, l_min: float, l_max: float, r_min: float, r_max: float,
                          lr_split: float = 0.5) -> T:
    from . import util
    size = section.size(0)
    if util.sample_uniform(0.0, 1.0) < lr_split:
        x = int((util.sample_uniform(l_min, l_max) * size) + 0.5)
        new_size = max(2, size - x)
    else:
        x = int((util.sample_uniform(r_min, r_max) * size) + 0.5)
        new_size = size + x
    return util.linear_interpolate_last_dim(section.contiguous(), new_size, align_corners=True)


def _corner_indices(corners: T) -> List[int]:
    return [int(c) for c in (corners.view(-1) == 1).nonzero(as_tuple=True)[0].tolist()]


def make_quasi_periodic(mod_sig: T, l_min: float = 0.2, l_max: float = 0.2, r_min: float = 0.2,
                        r_max: float = 0.2, lr_split: float = 0.5) -> T:
    """modulations.py:121-160: time-stretch every corner-to-corner section by a random amount."""
    from . import util
    assert mod_sig.ndim == 1
    top, bot = find_corners(mod_sig.unsqueeze(0))
    corners = top if float(top.sum()) > float(bot.sum()) else bot
    idx = _corner_indices(corners)
    if len(idx) < 2:
        return mod_sig
    sections, total, prev = [], 0, 0
    for c in idx:
        new_section = _time_stretch_section(mod_sig[prev:c + 1], l_min, l_max, r_min, r_max, lr_split)[:-1]
        total += new_section.size(0)
        sections.append(new_section)
        prev = c
    n = mod_sig.size(0)
    tail = mod_sig[prev:n]
    total += tail.size(0)
    if total < n:
        tail = util.linear_interpolate_last_dim(tail.contiguous(), tail.size(0) + (n - total), align_corners=True)
    sections.append(tail)
    return torch.cat(sections, dim=0)[:n]


def make_combined_mod_sig(n_samples: int, sr: float, freq: float, phase: float, shapes: List[str],
                          device=None) -> T:
    """modulations.py:191-210: a new random shape between every pair of bottom corners."""
    from . import util
    mod_sig = make_mod_signal(n_samples, sr, freq, phase, shape=util.choice(shapes), device=device)
    _, bot = find_corners(mod_sig.unsqueeze(0))
    idx = _corner_indices(bot)
    for a, b in zip(idx[:-1], idx[1:]):
        n = b - a + 1
        mod_sig[a:b + 1] = make_mod_signal(n, n, freq=1.0, phase=0.0, shape=util.choice(shapes), device=mod_sig.device)
    return mod_sig


def make_concave_convex_mod_sig(n_samples: int, sr: float, freq: float, phase: float = 0.0,
                                concave_min: float = 0.2, concave_max: float = 1.0, convex_min: float = 1.0,
                                convex_max: float = 3.0, concave_prob: float = 0.5, device=None) -> T:
    """modulations.py:163-188: triangle LFO with a random exponent per monotone segment."""
    from . import util
    mod_sig = make_mod_signal(n_samples, sr, freq, phase, shape="tri", device=device)
    top, bot = find_corners(mod_sig.unsqueeze(0))
    idx = _corner_indices(top + bot) + [mod_sig.size(0)]
    exp = torch.ones_like(mod_sig)
    prev = 0
    for c in idx:
        if util.sample_uniform(0.0, 1.0) < concave_prob:
            v = util.sample_uniform(concave_min, concave_max)
        else:
            v = util.sample_uniform(convex_min, convex_max)
        exp[prev:c] = v
        prev = c
    return mod_sig ** exp
