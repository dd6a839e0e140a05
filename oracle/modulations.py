"""Oracle restatement of mod_extraction/modulations.py (TEST INFRASTRUCTURE ONLY).

Pinned bit-for-bit against the reference's importable ``mod_extraction.modulations`` by
``tests/golden/make_golden.py`` (fixtures ``tests/golden/lfo.npz``, ``corners.npz``).

Bit-exactness notes (each one probed against torch 2.10 CPU in the build container):

* LFO phase (modulations.py:31).  ``cumsum`` of a constant fp32 ``step`` accumulates in fp64 on
  the CPU and rounds every output to fp32, and ``k*step`` is exact in fp64 for k < 2**29, so
      arg[k] = fl32( fl64(k+1) * fl64(step) ) + fl32(phase),   step = fl32(fl32(2pi)*fl32(f)) / fl32(sr)
  is a closed form (no scan).  ``arg[0]`` is one step, not zero.
* ``saw = remainder(arg, 2pi) / 2pi`` (modulations.py:32): fp32 fmod (exact) + sign fix, then a
  true fp32 division by fl32(2pi).
* transcendental values (cos, pow) come from torch's CPU kernels here; device libm differs in the
  last ulp, which is why LFO *values* get a 1e-5 tolerance while phase/saw/tri get bit equality.
* moving average (modulations.py:359-363): for window <= 8 torch's mean over the unfolded view is
  the left-to-right fp32 sum divided by the window (probed exactly equal for k=4, 8).
"""
import math
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import util

F32 = np.float32
TWO_PI_F32 = F32(2.0 * math.pi)
SHAPES = ("cos", "rect_cos", "inv_rect_cos", "tri", "saw", "rsaw", "sqr")


# ---------------------------------------------------------------------------------------------
# a1: LFO synthesis  (modulations.py:16-57)
# ---------------------------------------------------------------------------------------------
def lfo_step(sr: float, freq: float) -> np.float32:
    """fp32 phase increment exactly as ``2 * tr.pi * tr.full((n,), freq) / sr`` evaluates it."""
    return F32(F32(TWO_PI_F32 * F32(freq)) / F32(sr))


def lfo_argument(n_samples: int, sr: float, freq: float, phase: float) -> np.ndarray:
    k = np.arange(1, n_samples + 1, dtype=np.float64)
    running = (k * np.float64(lfo_step(sr, freq))).astype(F32)   # == torch.cumsum on CPU
    return (running + F32(phase)).astype(F32)


def lfo_saw(argument: np.ndarray) -> np.ndarray:
    m = np.fmod(argument, TWO_PI_F32).astype(F32)
    m = np.where((m != 0) & (m < 0), m + TWO_PI_F32, m).astype(F32)  # torch.remainder sign rule
    return (m / TWO_PI_F32).astype(F32)


def make_mod_signal(n_samples: int, sr: float, freq: float, phase: float = 0.0,
                    shape: str = "cos", exp: float = 1.0) -> torch.Tensor:
    assert n_samples > 0 and 0.0 < freq < sr / 2.0
    assert -2 * math.pi <= phase <= 2 * math.pi and exp > 0
    if shape not in SHAPES:
        raise ValueError("Unsupported shape")
    if shape in ("rect_cos", "inv_rect_cos"):       # rectified cosines run at half rate / phase
        freq, phase = freq / 2.0, phase / 2.0
    arg = lfo_argument(n_samples, sr, freq, phase)
    saw = lfo_saw(arg)
    t_arg = torch.from_numpy(arg)
    if shape == "cos":
        out = ((torch.cos(t_arg + math.pi) + 1.0) / 2.0).numpy()
    elif shape == "rect_cos":
        out = torch.abs(torch.cos(t_arg + (math.pi / 2.0))).numpy()
    elif shape == "inv_rect_cos":
        out = (-torch.abs(torch.cos(t_arg)) + 1.0).numpy()
    elif shape == "sqr":
        out = ((torch.sign(torch.cos(t_arg + math.pi)) + 1.0) / 2.0).numpy()
    elif shape == "saw":
        out = saw
    elif shape == "rsaw":
        out = (F32(1.0) - saw).astype(F32)
    else:  # tri
        tri = (F32(2.0) * saw).astype(F32)
        out = np.where(tri > F32(1.0), (F32(2.0) - tri).astype(F32), tri)
    res = torch.from_numpy(np.ascontiguousarray(out, dtype=F32))
    if exp != 1.0:
        res = res ** exp
    return res


# a13: baseline random LFOs (modulations.py:60-101); RNG draw order is part of the contract
def make_rand_mod_signal(batch_size: int, n_samples: int, sr: float, freq_min: float, freq_max: float,
                         shapes_gt: Optional[Sequence[str]] = None, shapes: Optional[List[str]] = None,
                         phase_gt: Optional[torch.Tensor] = None, phase_error: float = 0.5,
                         freq_gt: Optional[torch.Tensor] = None, freq_error: float = 0.25) -> torch.Tensor:
    if shapes is None:
        shapes = ["cos", "tri", "rect_cos", "inv_rect_cos", "saw", "rsaw"]
    rows = []
    for i in range(batch_size):
        if phase_gt is None:
            phase = util.sample_uniform(0.0, 2 * math.pi)
        else:
            phase = phase_gt[i]
            if phase_error > 0:
                phase += util.sample_uniform(-1.0, 1.0) * math.pi * phase_error   # in-place on the gt row,
                phase = (phase + (2 * math.pi)) % (2 * math.pi)                  # as the reference does
        if freq_gt is None:
            freq = util.sample_uniform(freq_min, freq_max)
        else:
            freq = freq_gt[i]
            if freq_error > 0:
                freq *= util.sample_uniform(1.0 - freq_error, 1.0 + freq_error)
                freq = torch.clip(freq, freq_min, freq_max)
        shape = util.choice(shapes) if shapes_gt is None else shapes_gt[i]
        rows.append(make_mod_signal(n_samples, sr, float(freq), float(phase), shape))
    return torch.stack(rows, dim=0)


# ---------------------------------------------------------------------------------------------
# a9: corner bookkeeping (modulations.py:219-363) -- all fp32, bit-exact
# ---------------------------------------------------------------------------------------------
def smoothen_np(x: np.ndarray, k: int) -> np.ndarray:
    x = np.asarray(x, dtype=F32)
    if k <= 1:
        return x
    n = x.shape[-1] - k + 1
    acc = np.zeros(x.shape[:-1] + (n,), dtype=F32)
    for j in range(k):
        acc = (acc + x[..., j:j + n]).astype(F32)
    return (acc / F32(k)).astype(F32)


def smoothen(x: torch.Tensor, smooth_n_frames: int) -> torch.Tensor:
    if smooth_n_frames <= 1:
        return x
    if smooth_n_frames <= 8:
        return torch.from_numpy(smoothen_np(x.numpy(), smooth_n_frames))
    return x.unfold(-1, smooth_n_frames, 1).mean(-1)     # torch's own order beyond the probed range


def find_corners_np(mod_sig: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """modulations.py:219-238.  Returns float32 0/1 maps (the reference returns float tensors)."""
    m = np.asarray(mod_sig, dtype=F32)
    assert m.ndim == 2
    d = (m[:, 1:] - m[:, :-1]).astype(F32)
    d_l, d_r = d[:, :-1], d[:, 1:]
    nudged = (d_r + F32(1e-16)).astype(F32)
    rising = np.where(d_l > 0, d_l, F32(0)).astype(F32)
    falling = np.where(d_l < 0, d_l, F32(0)).astype(F32)
    top = np.zeros_like(m)
    bot = np.zeros_like(m)
    top[:, 1:-1] = (-np.floor((rising * nudged).astype(F32))).astype(np.int64)
    bot[:, 1:-1] = (-np.floor((falling * nudged).astype(F32))).astype(np.int64)
    return top, bot


def find_corners(mod_sig: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    t, b = find_corners_np(mod_sig.numpy())
    return torch.from_numpy(t), torch.from_numpy(b)


def _stretch_one(m: np.ndarray, top: np.ndarray, bot: np.ndarray) -> np.ndarray:
    """modulations.py:260-291 for one LFO: rescale every monotone segment so that peaks land on
    1.0 and troughs on 0.0; the tail segment is re-anchored on the original last value."""
    n = m.shape[0]
    anchors = [(int(i), F32(1.0)) for i in np.nonzero(top == 1)[0]]
    anchors += [(int(i), F32(0.0)) for i in np.nonzero(bot == 1)[0]]
    anchors.append((n - 1, m[-1]))
    anchors.sort(key=lambda a: a[0])
    out = m.copy()
    prev_i, prev_target = 0, m[0]
    with np.errstate(divide="ignore", invalid="ignore"):
        for cur_i, target in anchors:
            if prev_target != target:
                have = F32(abs(F32(m[prev_i] - m[cur_i])))
                want = F32(abs(F32(prev_target - target)))
                gain = F32(want / have)
                seg = out[prev_i + 1:cur_i + 1]
                if seg.size:                       # torch would raise on an empty .min(); corners are >=1 apart
                    seg -= seg.min()
                    seg *= gain
                    seg += F32(target - seg[-1])
            prev_i, prev_target = cur_i, target
    return out


def stretch_corners(mod_sig: torch.Tensor, max_n_corners: int = 10, smooth_n_frames: int = 32) -> torch.Tensor:
    """modulations.py:294-307."""
    assert mod_sig.ndim == 2
    m = smoothen(mod_sig, smooth_n_frames).numpy().astype(F32)
    top, bot = find_corners_np(m)
    rows = []
    for r, t, b in zip(m, top, bot):
        rows.append(r if (t.sum() + b.sum()) > max_n_corners else _stretch_one(r, t, b))
    return torch.from_numpy(np.stack(rows, axis=0))


def stretch_corners_torch(mod_sig: torch.Tensor, max_n_corners: int = 10, smooth_n_frames: int = 32) -> torch.Tensor:
    """modulations.py:260-307 as differentiable torch code: the reference's operations in its order, but OUT of place.
    The reference's own formulation (``segment -= segment.min(); segment *= scale; segment += ...`` on a view of a clone)
    cannot be back-propagated: torch raises "one of the variables needed for gradient computation has been modified by an
    inplace operation" (``min`` saves the segment it is then subtracted from) -- checked here with torch 2.10 -- so the
    reference itself cannot train an unfrozen LFO model with should_stretch (lightning.py:258,294-296) as soon as one row is
    stretched.  This restatement defines the gradient of the SAME function for the tests of mx_stretch_corners_bwd; its
    values equal ``stretch_corners`` above."""
    assert mod_sig.ndim == 2
    if smooth_n_frames > 1:                                         # modulations.py:359-363
        mod_sig = mod_sig.unfold(dimension=-1, size=smooth_n_frames, step=1).mean(dim=-1)
    top_all, bot_all = find_corners(mod_sig.detach())
    rows = []
    for m, t, b in zip(mod_sig, top_all, bot_all):
        if t.sum() + b.sum() > max_n_corners:                       # modulations.py:301-303
            rows.append(m)
            continue
        idx = [(int(i), 1.0) for i in (t == 1).nonzero(as_tuple=True)[0]] + [(int(i), 0.0) for i in (b == 1).nonzero(as_tuple=True)[0]]
        idx += [(m.size(0) - 1, m[-1])]
        idx.sort(key=lambda v: v[0])
        prev_i, prev_anchor = 0, m[0]
        pieces = [m[:1]]
        for cur_i, target in idx:                                   # modulations.py:273-289
            segment = m[prev_i + 1:cur_i + 1]
            cur_val, orig_prev = m[cur_i], m[prev_i]
            if prev_anchor != target:
                scale = abs(prev_anchor - target) / abs(orig_prev - cur_val)
                segment = segment - segment.min()
                segment = segment * scale
                segment = segment + (target - segment[-1])
            pieces.append(segment)
            prev_i, prev_anchor = cur_i, target
        rows.append(torch.cat(pieces))
    return torch.stack(rows, dim=0)


def check_mod_sig_np(m: np.ndarray, top: np.ndarray, bot: np.ndarray,
                     min_top: int = 1, max_top: int = 6, min_bot: int = 1, max_bot: int = 6,
                     min_fraction_between_corners: float = 0.10) -> bool:
    """modulations.py:311-343."""
    n_top, n_bot = top.sum(), bot.sum()
    if n_top < min_top or n_bot < min_bot or n_top > max_top or n_bot > max_bot:
        return False
    min_gap = int(min_fraction_between_corners * m.shape[0])
    for c in (top, bot):
        idx = np.nonzero(c == 1)[0]
        if idx.size > 1 and np.diff(idx).min() < min_gap:
            return False
    return True


def find_valid_mod_sig_indices(mod_sig: torch.Tensor) -> List[int]:
    """modulations.py:346-356."""
    m = mod_sig.numpy().astype(F32)
    top, bot = find_corners_np(m)
    return [i for i in range(m.shape[0]) if check_mod_sig_np(m[i], top[i], bot[i])]


# ---------------------------------------------------------------------------------------------
# (f) rank-2 evaluation LFO variants (modulations.py:104-210); RNG order is part of the contract
# ---------------------------------------------------------------------------------------------
def _time_stretch_section(section: torch.Tensor, l_min, l_max, r_min, r_max, lr_split=0.5) -> torch.Tensor:
    size = section.size(0)
    if util.sample_uniform(0.0, 1.0) < lr_split:
        new_size = max(2, size - int((util.sample_uniform(l_min, l_max) * size) + 0.5))
    else:
        new_size = size + int((util.sample_uniform(r_min, r_max) * size) + 0.5)
    return util.linear_interpolate_last_dim(section, new_size)


def make_quasi_periodic(mod_sig: torch.Tensor, l_min=0.2, l_max=0.2, r_min=0.2, r_max=0.2,
                        lr_split=0.5) -> torch.Tensor:
    assert mod_sig.ndim == 1
    top, bot = find_corners(mod_sig.unsqueeze(0))
    corners = (top if top.sum() > bot.sum() else bot).squeeze(0)
    cidx = [int(c) for c in (corners == 1).nonzero(as_tuple=True)[0]]
    if len(cidx) < 2:
        return mod_sig
    pieces, total, prev = [], 0, 0
    for c in cidx:
        piece = _time_stretch_section(mod_sig[prev:c + 1], l_min, l_max, r_min, r_max, lr_split)[:-1]
        total += piece.size(0)
        pieces.append(piece)
        prev = c
    n = mod_sig.size(0)
    tail = mod_sig[prev:n]
    total += tail.size(0)
    if total < n:
        tail = util.linear_interpolate_last_dim(tail, tail.size(0) + (n - total))
    pieces.append(tail)
    return torch.cat(pieces, dim=0)[:n]


def make_combined_mod_sig(n_samples: int, sr: float, freq: float, phase: float, shapes: List[str]) -> torch.Tensor:
    sig = make_mod_signal(n_samples, sr, freq, phase, shape=util.choice(shapes))
    _, bot = find_corners(sig.unsqueeze(0))
    cidx = [int(c) for c in (bot.squeeze(0) == 1).nonzero(as_tuple=True)[0]]
    for a, b in zip(cidx[:-1], cidx[1:]):
        length = b - a + 1
        sig[a:b + 1] = make_mod_signal(length, length, freq=1.0, phase=0.0, shape=util.choice(shapes))
    return sig


def make_concave_convex_mod_sig(n_samples: int, sr: float, freq: float, phase: float = 0.0, concave_min: float = 0.2,
                                concave_max: float = 1.0, convex_min: float = 1.0, convex_max: float = 3.0,
                                concave_prob: float = 0.5) -> torch.Tensor:
    """modulations.py:163-188: a triangle whose monotone segments are each raised to a random exponent
    (two host draws per segment: concave or convex, then the exponent)."""
    sig = make_mod_signal(n_samples, sr, freq, phase, shape="tri")
    top, bot = find_corners(sig.unsqueeze(0))
    ends = [int(c) for c in ((top + bot).squeeze(0) == 1).nonzero(as_tuple=True)[0]] + [sig.size(0)]
    power = torch.ones_like(sig)
    start = 0
    for end in ends:
        lo, hi = (concave_min, concave_max) if util.sample_uniform(0.0, 1.0) < concave_prob else (convex_min, convex_max)
        power[start:end] = util.sample_uniform(lo, hi)
        start = end
    return sig ** power
