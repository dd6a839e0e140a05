"""Host mirror of mod_extraction/models.py: same classes, constructor arguments, forward signatures
and state-dict keys; every forward/backward runs hand-written HIP kernels through the C ABI.

* ``Spectral2DCNN`` (models.py:128-215): log-mel front end (``mx_logmel_fwd``) and six fused
  LayerNorm -> Conv2d(5x13) -> MaxPool(2,1) -> PReLU blocks on the fp32 matrix cores
  (``mx_conv_block_{fwd,dgrad,wgrad}``, ``mx_plane_stats``, ``mx_ln_prelu_bwd``), head
  (``mx_head_{fwd,bwd}``), wired into autograd by one ``torch.autograd.Function``.
  The ``torch.nn`` layer objects inside ``self.cnn`` / ``self.output`` are parameter holders only
  (they give the reference's state-dict keys and default initialisation); they are never called.
* ``LSTMEffectModel`` / ``HiddenStateModel`` (models.py:292-339) and ``RandomLFO`` (models.py:19-69).
"""
import logging
import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch import Tensor as T, nn

from . import _hip

log = logging.getLogger(__name__)

import os as _os

# Arithmetic of the 64->64 channel convolutions (forward + data gradient):
#   "f16x3"  fp16 matrix cores on split operands (hi + lo pairs, 3 MFMAs per product group) -- fp32-equivalent
#            accuracy (csrc/conv_f16.hip), 5.3x the fp32 MFMA rate;   "f32"  exact fp32 MFMA (csrc/conv2d.hip).
CONV_PRECISION = _os.environ.get("MODEX_CONV_PRECISION", "f16x3")
DEBUG_TAP = None       # set to a dict to capture backward intermediates (tools/debug_cnn_bwd.py)
PITCH = 352            # activation row pitch (floats); 345 frames + pad (csrc/conv_common.h)
LN_EPS = 1e-5          # torch.nn.LayerNorm default


# ---------------------------------------------------------------------------------------------
# mel front-end constants (torchaudio 0.13.1 MelSpectrogram defaults as used at models.py:170-175)
# ---------------------------------------------------------------------------------------------
def htk_mel_filterbank(n_freqs: int, f_min: float, f_max: float, n_mels: int, sample_rate: int) -> T:
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale='htk'): triangular filters."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + (f_min / 700.0))
    m_max = 2595.0 * math.log10(1.0 + (f_max / 700.0))
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down_slopes = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up_slopes = slopes[:, 2:] / f_diff[1:]
    return torch.max(torch.zeros(1), torch.min(down_slopes, up_slopes))


def _band_limits(fb: T) -> Tuple[T, T]:
    nz = fb != 0
    any_nz = nz.any(dim=0)
    idx = torch.arange(fb.size(0)).unsqueeze(1).expand_as(fb)
    lo = torch.where(nz, idx, torch.full_like(idx, fb.size(0))).min(dim=0).values
    hi = torch.where(nz, idx + 1, torch.zeros_like(idx)).max(dim=0).values
    lo = torch.where(any_nz, lo, torch.zeros_like(lo))
    return lo.to(torch.int32), hi.to(torch.int32)


class _Buffers(nn.Module):
    """Plain namespace module (gives buffers their torchaudio state-dict key prefixes)."""


class MelSpectrogramHIP(nn.Module):
    """State-dict compatible stand-in for torchaudio's MelSpectrogram (keys
    ``spectrogram.window`` and ``mel_scale.fb``); the transform itself is the HIP kernel."""

    def __init__(self, sample_rate: int, n_fft: int, hop_length: int, n_mels: int) -> None:
        super().__init__()
        if n_fft not in (512, 1024, 2048):
            raise NotImplementedError("mx_logmel_fwd is built for n_fft in {512, 1024, 2048}")
        self.sample_rate, self.n_fft, self.hop_length, self.n_mels = sample_rate, n_fft, hop_length, n_mels
        self.spectrogram = _Buffers()
        self.spectrogram.register_buffer("window", torch.hann_window(n_fft))
        self.mel_scale = _Buffers()
        self.mel_scale.register_buffer("fb", htk_mel_filterbank(n_fft // 2 + 1, 0.0, float(sample_rate // 2),
                                                                n_mels, sample_rate))
        k = torch.arange(n_fft, dtype=torch.float64) * (-2.0 * math.pi / n_fft)
        self.register_buffer("twiddle", torch.stack([torch.cos(k), torch.sin(k)], dim=1).float(), persistent=False)
        self._bands: Optional[Tuple[T, T, int]] = None

    def bands(self) -> Tuple[T, T]:
        fb = self.mel_scale.fb
        if self._bands is None or self._bands[2] != fb._version or self._bands[0].device != fb.device:
            lo, hi = _band_limits(fb.detach().cpu())
            self._bands = (lo.to(fb.device), hi.to(fb.device), fb._version)
        return self._bands[0], self._bands[1]

    def log_mel(self, x: T, n_frames: int, eps: float, masks: Sequence[int] = (0, 0, 0, 0), pitch: int = PITCH) -> T:
        """x (B, C, N) -> (B, C, n_mels, pitch) = log(clip(mel, eps)) with SpecAugment ranges."""
        B, C, N = x.shape
        xc = x.contiguous().float()
        out = torch.empty((B, C, self.n_mels, pitch), device=x.device, dtype=torch.float32)
        lo, hi = self.bands()
        f0, f1, t0, t1 = (int(v) for v in masks)
        _hip.call("mx_logmel_fwd", _hip.ptr(xc), B * C, N, _hip.ptr(self.spectrogram.window), _hip.ptr(self.twiddle),
                  _hip.ptr(self.mel_scale.fb), _hip.ptr(lo), _hip.ptr(hi), self.n_fft, self.hop_length, self.n_mels,
                  n_frames, pitch, float(eps), f0, f1, t0, t1, _hip.ptr(out), _hip.stream())
        return out


def specaugment_bounds(size: int, mask_param: int) -> Tuple[int, int]:
    """torchaudio.functional.mask_along_axis: one mask per batch; two host ``torch.rand(1)`` draws
    (value, then min_value); masked range [int(min_value), int(min_value) + int(value))."""
    value = torch.rand(1) * mask_param
    min_value = torch.rand(1) * (size - value)
    start = int(min_value.long())
    return start, start + int(value.long())


# ---------------------------------------------------------------------------------------------
# the CNN stack as one autograd node
# ---------------------------------------------------------------------------------------------
def _pack(w: T, flip: int) -> T:
    out = torch.empty(w.numel(), device=w.device, dtype=torch.float32)
    _hip.call("mx_conv_pack_weights", _hip.ptr(w.contiguous()), w.size(0), w.size(1), flip, _hip.ptr(out),
              _hip.stream())
    return out


def _pack_f16(w: T, flip: int) -> Tuple[T, T]:
    n = 4 * 5 * 13 * 64 * 16
    hi = torch.empty(n, device=w.device, dtype=torch.float16)
    lo = torch.empty(n, device=w.device, dtype=torch.float16)
    _hip.call("mx_conv_pack_weights_f16", _hip.ptr(w.contiguous()), flip, _hip.ptr(hi), _hip.ptr(lo), _hip.stream())
    return hi, lo


# first block on the fp16 pipes too (MODEX_BLOCK1=f32 keeps it on the exact-fp32 MFMA kernel; A/B knob)
BLOCK1_F16 = os.environ.get("MODEX_BLOCK1", "f16x3") != "f32"
# weight gradient of the 64-channel blocks: sparse (2:4 along the pooling pair) or dense matrix instruction
WGRAD_SPARSE = os.environ.get("MODEX_WGRAD", "sparse") != "dense"
WGRAD_SPARSE_MAX_T = int(os.environ.get("MODEX_WGRAD_SP_MAXT", "4"))     # dilations above it: dense kernel (no shared fragment blocks)
DGRAD_SPARSE = os.environ.get("MODEX_DGRAD", "sparse") != "dense"
DIRECT_GRADS = os.environ.get("MODEX_DIRECT_GRADS", "1") != "0"   # parameter gradients written straight into FlatAdamW's flat buffer
STATS_FUSED = os.environ.get("MODEX_STATS", "fused") != "sweep"   # next block's LayerNorm statistics from the forward epilogue
# LayerNorm / PReLU backward written straight into the pooled operand of the block below (blocks whose two gradients both run
# on the sparse instruction): dL/dp never exists in fp32.  "split" keeps the two passes (A/B knob).
GPOOL_FUSED = os.environ.get("MODEX_GPOOL", "fused") != "split"
LN_FUSED = os.environ.get("MODEX_LN", "fused") != "sweep"      # LayerNorm-backward statistics from the data-gradient epilogue
# the gradient of the first block handed to its weight gradient as f16x3 pairs (written in place by the LayerNorm backward, scale
# from the bound on max|G|) instead of fp32 values that the weight gradient scales / splits while staging ("0": round-4 route)
BLOCK1_PAIR = os.environ.get("MODEX_BLOCK1_PAIR", "1") != "0"


def _use_f16(cin: int, precision: str) -> bool:
    return precision == "f16x3" and cin == 64


def _reduce_rows(part: T, rows: int, cols: int, out: Optional[T] = None) -> T:
    if out is None:
        out = torch.empty(cols, device=part.device, dtype=torch.float32)
    _hip.call("mx_reduce_rows", _hip.ptr(part), rows, cols, 0, _hip.ptr(out), _hip.stream())
    return out


def _direct_grad_views(params) -> Optional[List[T]]:
    """The parameters' ``.grad`` tensors when the backward pass may write its results straight into them: all of them are
    contiguous views of ONE flat gradient buffer (optim.FlatAdamW) and the caller has armed the in-place path for THIS
    backward (``with optimizer.direct_backward(): loss.backward()`` -- ``_modex_fresh`` on the buffer, set on entry, consumed
    here, cleared on exit; ``Trainer.train_step`` does it for the first backward after ``zero_grad()``).  autograd then gets
    ``None`` for those inputs and skips its 20 ``grad += g`` launches per step.  Any other situation -- no flat buffer, a
    backward outside such a scope (weight penalties, a second sub-batch, ``torch.autograd.grad``), plain ``torch.optim`` --
    keeps the ordinary accumulate path, so nothing already in ``.grad`` is ever overwritten silently."""
    views = []
    base = None
    for p in params:
        if not (p.is_leaf and p.requires_grad):          # (.grad of a non-leaf tensor warns)
            return None
        g = p.grad
        if g is None or not g.is_contiguous() or g._base is None or g.dtype != torch.float32:
            return None
        if base is None:
            base = g._base
        elif g._base is not base:
            return None
        views.append(g)
    if base is None or not getattr(base, "_modex_fresh", False):
        return None
    base._modex_fresh = False
    return views


def _pooled_only(l: int, cin: int, dilations, precision: str, n_frames: int) -> bool:
    """Block l (0-based) consumes its gradient only as the pooled channels-last pair: f16x3, both gradients sparse."""
    return (l > 0 and _use_f16(cin, precision) and WGRAD_SPARSE and DGRAD_SPARSE
            and int(dilations[l]) <= WGRAD_SPARSE_MAX_T and n_frames <= PITCH - 1)


class _CNNStack(torch.autograd.Function):
    """logmel (B,Cin,H,PITCH) -> (sigmoid output (B,L,W), latent (B,64,W)).
    params: [w1,b1,a1, ..., w6,b6,a6, wout, bout]."""

    @staticmethod
    def forward(ctx, logmel: T, n_frames: int, dilations: Tuple[int, ...], precision: str, *params: T):
        n_blocks = len(dilations)
        B, cin, H, _ = logmel.shape
        dev = logmel.device
        st = _hip.stream()
        cur, slope = logmel, None
        stats_part = None                 # {sum, sum of squares} per pooled row left by the previous block's forward epilogue
        prev_bias = None                  # ... taken of PReLU(out) - PReLU(bias): the finish pass needs that bias
        saved: List[T] = []
        # the fp16 operand pairs of the 64-channel blocks are kept for the weight gradient when a backward pass
        # will follow (5.9 GB at 256 clips x 2 s: cheaper than re-deriving them from the saved activations)
        keep_splits = any(ctx.needs_input_grad)
        ctx.splits = {}
        for l in range(n_blocks):
            w, b, a = params[3 * l], params[3 * l + 1], params[3 * l + 2]
            stats = torch.empty((B, cin, 2), device=dev, dtype=torch.float32)
            if stats_part is not None:
                _hip.call("mx_plane_stats_finish", _hip.ptr(stats_part), _hip.ptr(prev_bias), _hip.ptr(slope), B, cin, H,
                          n_frames, LN_EPS, _hip.ptr(stats), st)
            else:
                _hip.call("mx_plane_stats", _hip.ptr(cur), _hip.ptr(slope), B, cin, H, n_frames, LN_EPS,
                          _hip.ptr(stats), st)
            stats_part = None
            fuse_stats = STATS_FUSED and l + 1 < n_blocks          # (the head takes the last block's output as it is)
            a_out = a.contiguous()
            p = torch.empty((B, 64, H // 2, PITCH), device=dev, dtype=torch.float32)
            amax = torch.empty((B, 64, H // 2, PITCH), device=dev, dtype=torch.uint8)
            if _use_f16(cin, precision):
                x_hi = torch.empty((B, H, 4, PITCH, 16), device=dev, dtype=torch.float16)
                x_lo = torch.empty((B, H, 4, PITCH, 16), device=dev, dtype=torch.float16)
                _hip.call("mx_conv_prep_fwd_f16", _hip.ptr(cur), _hip.ptr(stats), _hip.ptr(slope), B, H, n_frames,
                          _hip.ptr(x_hi), _hip.ptr(x_lo), st)
                w_hi, w_lo = _pack_f16(w, 0)
                if fuse_stats:
                    stats_part = torch.empty((B, H // 2, 64, 2), device=dev, dtype=torch.float32)
                _hip.call("mx_conv_block_fwd_f16", _hip.ptr(x_hi), _hip.ptr(x_lo), _hip.ptr(w_hi), _hip.ptr(w_lo),
                          _hip.ptr(b.contiguous()), B, H, n_frames, int(dilations[l]), _hip.ptr(p), _hip.ptr(amax),
                          _hip.ptr(a_out) if fuse_stats else None, _hip.ptr(stats_part), st)
                if keep_splits:
                    ctx.splits[l] = (x_hi, x_lo)
                del x_hi, x_lo
            elif precision == "f16x3" and l == 0 and cin == 2 and BLOCK1_F16 and int(dilations[0]) == 1:
                # first block: (kernel row, channel) pairs are the operand's 16 channels; one K stage (the kernel is built for
                # the undilated first block of every shipped config; a dilated one takes the exact-fp32 kernel below)
                xk_hi = torch.empty((B, H, 1, PITCH, 16), device=dev, dtype=torch.float16)
                xk_lo = torch.empty((B, H, 1, PITCH, 16), device=dev, dtype=torch.float16)
                _hip.call("mx_conv_prep_fwd_kvec_f16", _hip.ptr(cur), _hip.ptr(stats), B, H, n_frames, _hip.ptr(xk_hi),
                          _hip.ptr(xk_lo), st)
                wk_hi = torch.empty(13 * 2 * 64 * 8, device=dev, dtype=torch.float16)
                wk_lo = torch.empty(13 * 2 * 64 * 8, device=dev, dtype=torch.float16)
                _hip.call("mx_conv_pack_weights_kvec_f16", _hip.ptr(w.detach().contiguous()), _hip.ptr(wk_hi),
                          _hip.ptr(wk_lo), st)
                if fuse_stats:
                    stats_part = torch.empty((B, H // 2, 64, 2), device=dev, dtype=torch.float32)
                _hip.call("mx_conv_block1_fwd_f16", _hip.ptr(xk_hi), _hip.ptr(xk_lo), _hip.ptr(wk_hi), _hip.ptr(wk_lo),
                          _hip.ptr(b.contiguous()), B, H, n_frames, _hip.ptr(p), _hip.ptr(amax),
                          _hip.ptr(a_out) if fuse_stats else None, _hip.ptr(stats_part), st)
                if keep_splits:
                    ctx.splits[l] = (xk_hi, xk_lo)              # the weight gradient consumes the same operand
                del xk_hi, xk_lo
            else:
                wt = _pack(w, 0)
                _hip.call("mx_conv_block_fwd", _hip.ptr(cur), _hip.ptr(stats), _hip.ptr(slope), _hip.ptr(wt),
                          _hip.ptr(b.contiguous()), B, cin, H, n_frames, int(dilations[l]), 1 if l == 0 else 0,
                          _hip.ptr(p), _hip.ptr(amax), st)
            saved += [cur, stats, amax]
            cur, slope, cin, H, prev_bias = p, a_out, 64, H // 2, b.contiguous()
        wout, bout = params[3 * n_blocks], params[3 * n_blocks + 1]
        L = wout.size(0)
        latent = torch.empty((B, 64, n_frames), device=dev, dtype=torch.float32)
        out = torch.empty((B, L, n_frames), device=dev, dtype=torch.float32)
        _hip.call("mx_head_fwd", _hip.ptr(cur), _hip.ptr(slope), _hip.ptr(wout.contiguous()),
                  _hip.ptr(bout.contiguous()), B, 64, H, n_frames, L, _hip.ptr(latent), _hip.ptr(out), st)
        ctx.save_for_backward(*saved, cur, latent, out, *params)
        ctx.param_objs = params               # the Parameter objects themselves: their .grad views are looked up in backward
        ctx.meta = (n_frames, tuple(dilations), n_blocks, precision)
        return out, latent

    @staticmethod
    def backward(ctx, d_out: Optional[T], d_latent: Optional[T]):
        n_frames, dilations, n_blocks, precision = ctx.meta
        tensors = ctx.saved_tensors
        saved, p_last, latent, out = tensors[:3 * n_blocks], tensors[3 * n_blocks], tensors[3 * n_blocks + 1], \
            tensors[3 * n_blocks + 2]
        params = tensors[3 * n_blocks + 3:]
        dev = out.device
        st = _hip.stream()
        B, L = out.size(0), out.size(1)
        wout = params[3 * n_blocks]
        grads: List[Optional[T]] = [None] * len(params)
        direct = _direct_grad_views(ctx.param_objs) if DIRECT_GRADS else None

        def gout(i: int) -> Optional[T]:          # where parameter i's gradient is written: its .grad view, or a fresh tensor
            return direct[i].view(-1) if direct is not None else None
        if d_out is None:
            d_out = torch.zeros_like(out)
        d_out = d_out.contiguous()
        d_latent = d_latent.contiguous() if d_latent is not None else None
        Hl = p_last.size(2)
        G = torch.empty_like(p_last)
        dw_part = torch.empty((B, L * 64), device=dev, dtype=torch.float32)
        db_part = torch.empty((B, L), device=dev, dtype=torch.float32)
        ds_part = torch.empty((B, 64), device=dev, dtype=torch.float32)
        slope_last = params[3 * (n_blocks - 1) + 2].contiguous()
        # max|G| of the last block's gradient for its f16x3 scale: taken while G is written (no sweep)
        # one zeroed workspace for every atomic-max cell of this backward pass (each used to be its own fill launch): cells
        # 4 l .. 4 l + 3 belong to block l
        zws = torch.zeros(4 * (n_blocks + 1), device=dev, dtype=torch.int32)
        pair_scale, pair1 = None, False   # the first block's gradient as f16x3 pairs (BLOCK1_PAIR)
        gmax_ws = zws[4 * n_blocks:4 * n_blocks + 1] if _use_f16(saved[3 * (n_blocks - 1)].size(1), precision) else None
        _hip.call("mx_head_bwd", _hip.ptr(p_last), _hip.ptr(slope_last), _hip.ptr(wout.contiguous()),
                  _hip.ptr(latent), _hip.ptr(out), _hip.ptr(d_out), _hip.ptr(d_latent), B, 64, Hl, n_frames, L,
                  _hip.ptr(G), _hip.ptr(dw_part), _hip.ptr(db_part), _hip.ptr(ds_part), _hip.ptr(gmax_ws), st)
        grads[3 * n_blocks] = _reduce_rows(dw_part, B, L * 64, gout(3 * n_blocks)).view_as(wout)
        grads[3 * n_blocks + 1] = _reduce_rows(db_part, B, L, gout(3 * n_blocks + 1))
        grads[3 * (n_blocks - 1) + 2] = _reduce_rows(ds_part, B, 64, gout(3 * (n_blocks - 1) + 2))
        bsum = None                      # by-product of mx_ln_prelu_bwd for the block below: bias partials (gmax_ws: max|G| bits)
        pooled = None                    # (gc_hi, gc_lo, gc_idx, gidx, scale) left for the block below by the fused LN backward
        for l in range(n_blocks - 1, -1, -1):
            x_in, stats, amax = saved[3 * l], saved[3 * l + 1], saved[3 * l + 2]
            w = params[3 * l]
            cin, H = x_in.size(1), x_in.size(2)
            slope_prev = params[3 * (l - 1) + 2].contiguous() if l > 0 else None
            # bias gradient: sum of G over (b, h, w)
            if DEBUG_TAP is not None:
                if G is not None and pair_scale is None:   # (with the fused LayerNorm backward dL/dp of blocks 2-4 never exists in fp32)
                    DEBUG_TAP[f"G{l}"] = G.clone()
                DEBUG_TAP[f"amax{l}"] = amax.clone()
                DEBUG_TAP[f"p{l}"] = (p_last if l == n_blocks - 1 else saved[3 * (l + 1)]).clone()
            if bsum is None:
                assert G is not None
                bsum = torch.empty((B, 64), device=dev, dtype=torch.float32)
                _hip.call("mx_plane_sum", _hip.ptr(G), B * 64, H // 2, n_frames, _hip.ptr(bsum), st)
            grads[3 * l + 1] = _reduce_rows(bsum, B, 64, gout(3 * l + 1))
            bsum = None
            # weight gradient (+ data gradient below) -- f16x3 path: both share the prepared operand pairs
            rows = B * H
            f16 = _use_f16(cin, precision)
            dW = direct[3 * l] if direct is not None else torch.empty_like(w)
            if f16:
                dz_hi = dz_lo = None
                ready = gmax_ws is not None
                ws = gmax_ws if ready else torch.empty(1, device=dev, dtype=torch.int32)
                gmax_ws = None
                scale = torch.empty(2, device=dev, dtype=torch.float32)
                # sparse matrix instruction: the pooled gradient is the compressed operand, the argmax its index bits.
                # Weight gradient: dilations <= 4 (for >= 8 the taps share no fragment blocks: dense kernel on the routed
                # full-resolution pair); data gradient: every dilation.
                sparse = WGRAD_SPARSE and int(dilations[l]) <= WGRAD_SPARSE_MAX_T and n_frames <= PITCH - 1
                sparse_d = DGRAD_SPARSE and l > 0 and n_frames <= PITCH - 1
                gidx = gc_hi = gc_lo = gc_idx = None
                Hp = H // 2
                need_routed = (not sparse) or (l > 0 and not sparse_d)   # a dense kernel consumes the routed full-resolution pair
                if need_routed:
                    dz_hi = torch.empty((B, H, 4, PITCH, 16), device=dev, dtype=torch.float16)
                    dz_lo = torch.empty((B, H, 4, PITCH, 16), device=dev, dtype=torch.float16)
                if pooled is not None:
                    gc_hi, gc_lo, gc_idx, gidx, scale = pooled       # made by the LayerNorm backward of the block above
                    pooled = None
                else:
                    _hip.call("mx_conv_prep_dgrad_f16", _hip.ptr(G), _hip.ptr(amax), B, H, n_frames, _hip.ptr(ws),
                              1 if ready else 0, _hip.ptr(scale), _hip.ptr(dz_hi), _hip.ptr(dz_lo), st)
                if gc_hi is None and (sparse or sparse_d):
                    # one pass over G: the channels-last pooled pair both sparse kernels read, the data gradient's index
                    # words and (for the weight gradient) the planar ones
                    gc_hi = torch.empty((B, Hp, 4, PITCH, 16), device=dev, dtype=torch.float16)
                    gc_lo = torch.empty((B, Hp, 4, PITCH, 16), device=dev, dtype=torch.float16)
                    gc_idx = torch.empty((B, Hp, 4, PITCH), device=dev, dtype=torch.int32)
                    if sparse:
                        gidx = torch.empty((B, 64, Hp, 22, 2), device=dev, dtype=torch.int16)
                    _hip.call("mx_conv_prep_gpool_cl_f16", _hip.ptr(G), _hip.ptr(amax), _hip.ptr(scale), B, H, n_frames,
                              _hip.ptr(gc_hi), _hip.ptr(gc_lo), _hip.ptr(gc_idx), _hip.ptr(gidx), st)
                if l in ctx.splits:
                    x_hi, x_lo = ctx.splits.pop(l)
                else:
                    x_hi = torch.empty((B, H, 4, PITCH, 16), device=dev, dtype=torch.float16)
                    x_lo = torch.empty((B, H, 4, PITCH, 16), device=dev, dtype=torch.float16)
                    _hip.call("mx_conv_prep_fwd_f16", _hip.ptr(x_in), _hip.ptr(stats), _hip.ptr(slope_prev), B, H,
                              n_frames, _hip.ptr(x_hi), _hip.ptr(x_lo), st)
                if sparse:
                    prow = B * Hp
                    rps = max(1, -(-prow // 256))           # 256 slabs x 5 kernel rows = 5 full rounds of 256 workgroups
                    n_slabs = -(-prow // rps)
                    part = torch.empty(n_slabs * 65 * 64 * 64, device=dev, dtype=torch.float32)
                    _hip.call("mx_conv_block_wgrad_sp_f16", _hip.ptr(gc_hi), _hip.ptr(gc_lo), _hip.ptr(gidx), _hip.ptr(x_hi),
                              _hip.ptr(x_lo), _hip.ptr(scale), B, H, n_frames, int(dilations[l]), rps, _hip.ptr(part),
                              _hip.ptr(dW), st)
                    del gidx
                else:
                    rps = max(1, -(-rows // 256))            # 256 slabs x 5 kernel rows = 5 full rounds of 256 workgroups
                    n_slabs = -(-rows // rps)
                    part = torch.empty(n_slabs * 65 * 64 * 64, device=dev, dtype=torch.float32)
                    _hip.call("mx_conv_block_wgrad_f16", _hip.ptr(dz_hi), _hip.ptr(dz_lo), _hip.ptr(x_hi), _hip.ptr(x_lo),
                              _hip.ptr(scale), B, H, int(dilations[l]), rps, _hip.ptr(part), _hip.ptr(dW), st)
                del part
                if not (sparse_d and LN_FUSED):
                    del x_hi, x_lo
            elif l == 0 and 0 in ctx.splits and pair_scale is not None:
                # first block on the fp16 pipes, gradient already in f16x3 pairs (mx_ln_prelu_bwd_pair): routed while staging
                xk_hi, xk_lo = ctx.splits.pop(0)
                rps = max(1, -(-rows // 2048))
                n_slabs = -(-rows // rps)
                part = torch.empty(n_slabs * 13 * 64 * 16, device=dev, dtype=torch.float32)
                _hip.call("mx_conv_block1_wgrad_pair_f16", _hip.ptr(G), _hip.ptr(amax), _hip.ptr(pair_scale), _hip.ptr(xk_hi),
                          _hip.ptr(xk_lo), B, H, n_frames, rps, _hip.ptr(part), _hip.ptr(dW), st)
                pair_scale = None
                del part, xk_hi, xk_lo
            elif l == 0 and 0 in ctx.splits and gmax_ws is not None:
                # first block on the fp16 pipes: the kept k-vector operand, gradient routed / scaled / split on the fly
                xk_hi, xk_lo = ctx.splits.pop(0)
                rps = max(1, -(-rows // 2048))
                n_slabs = -(-rows // rps)
                part = torch.empty(n_slabs * 13 * 64 * 16, device=dev, dtype=torch.float32)
                scale1 = torch.empty(2, device=dev, dtype=torch.float32)
                _hip.call("mx_conv_block1_wgrad_f16", _hip.ptr(G), _hip.ptr(amax), _hip.ptr(gmax_ws), _hip.ptr(xk_hi),
                          _hip.ptr(xk_lo), B, H, n_frames, rps, _hip.ptr(scale1), _hip.ptr(part), _hip.ptr(dW), st)
                gmax_ws = None
                del part, xk_hi, xk_lo
            else:
                rps = max(1, -(-rows // (256 if cin == 64 else 1024)))
                n_slabs = -(-rows // rps)
                part = torch.empty(n_slabs * 65 * 64 * cin, device=dev, dtype=torch.float32)
                _hip.call("mx_conv_block_wgrad", _hip.ptr(G), _hip.ptr(amax), _hip.ptr(x_in), _hip.ptr(stats),
                          _hip.ptr(slope_prev), B, cin, H, n_frames, int(dilations[l]), rps, _hip.ptr(part),
                          _hip.ptr(dW), st)
                del part
            grads[3 * l] = dW
            if l > 0:
                dxhat = torch.empty((B, 64, H, PITCH), device=dev, dtype=torch.float32)
                ln_part, fuse_g, pair1 = None, False, False
                if f16 and sparse_d:
                    # sparse matrix instruction, transposed tiles: pooled channels-last gradient x fragment-packed weights
                    ws_hi = torch.empty(4 * 3 * 2 * 13 * 2 * 64 * 16, device=dev, dtype=torch.float16)
                    ws_lo = torch.empty(4 * 3 * 2 * 13 * 2 * 64 * 16, device=dev, dtype=torch.float16)
                    _hip.call("mx_conv_pack_weights_sp_f16", _hip.ptr(w.detach().contiguous()), _hip.ptr(ws_hi),
                              _hip.ptr(ws_lo), st)
                    if LN_FUSED:
                        # the epilogue also leaves the plane statistics of the LayerNorm backward below (x = xhat) and, when
                        # that pass writes the block below's pooled operand itself, max|dxhat| / max|xhat| for its scale
                        ln_part = torch.empty((B, 64, H, 2, 2), device=dev, dtype=torch.float32)
                        fuse_g = GPOOL_FUSED and _pooled_only(l - 1, saved[3 * (l - 1)].size(1), dilations, precision, n_frames)
                        pair1 = BLOCK1_PAIR and l == 1 and 0 in ctx.splits and not fuse_g
                        gx_bits = zws[4 * l:4 * l + 2] if (fuse_g or pair1) else None
                        _hip.call("mx_conv_block_dgrad_sp_f16", _hip.ptr(gc_hi), _hip.ptr(gc_lo), _hip.ptr(gc_idx),
                                  _hip.ptr(ws_hi), _hip.ptr(ws_lo), _hip.ptr(scale), B, H, n_frames, int(dilations[l]),
                                  _hip.ptr(dxhat), _hip.ptr(x_hi), _hip.ptr(x_lo), _hip.ptr(ln_part), _hip.ptr(gx_bits), st)
                        del x_hi, x_lo
                    else:
                        _hip.call("mx_conv_block_dgrad_sp_f16", _hip.ptr(gc_hi), _hip.ptr(gc_lo), _hip.ptr(gc_idx),
                                  _hip.ptr(ws_hi), _hip.ptr(ws_lo), _hip.ptr(scale), B, H, n_frames, int(dilations[l]),
                                  _hip.ptr(dxhat), None, None, None, None, st)
                    del gc_hi, gc_lo, gc_idx
                elif f16:
                    w_hi, w_lo = _pack_f16(w, 1)
                    _hip.call("mx_conv_block_dgrad_f16", _hip.ptr(dz_hi), _hip.ptr(dz_lo), _hip.ptr(w_hi), _hip.ptr(w_lo),
                              _hip.ptr(scale), B, H, n_frames, int(dilations[l]), _hip.ptr(dxhat), st)
                    del dz_hi, dz_lo
                else:
                    wt_f = _pack(w, 1)
                    _hip.call("mx_conv_block_dgrad", _hip.ptr(G), _hip.ptr(amax), _hip.ptr(wt_f), B, H, n_frames,
                              int(dilations[l]), _hip.ptr(dxhat), st)
                if DEBUG_TAP is not None:
                    DEBUG_TAP[f"dxhat{l}"] = dxhat.clone()
                ds_part = torch.empty((B, 64), device=dev, dtype=torch.float32)
                bsum = torch.empty((B, 64), device=dev, dtype=torch.float32)
                if fuse_g:
                    # dL/dp of the block below never exists in fp32: LayerNorm / PReLU backward -> scale from a bound on
                    # max|G| -> split -> channels-last pooled pair + index words, one pass (csrc/dgrad_sp_f16.hip)
                    m12 = torch.empty((B, 64, 2), device=dev, dtype=torch.float32)
                    bound_ws = torch.empty(1, device=dev, dtype=torch.int32)
                    scale_n = torch.empty(2, device=dev, dtype=torch.float32)
                    _hip.call("mx_ln_bwd_finish", _hip.ptr(ln_part), _hip.ptr(stats), _hip.ptr(slope_prev), _hip.ptr(gx_bits),
                              B, 64, H, n_frames, _hip.ptr(m12), _hip.ptr(bound_ws), _hip.ptr(scale_n), st)
                    gn_hi = torch.empty((B, H, 4, PITCH, 16), device=dev, dtype=torch.float16)
                    gn_lo = torch.empty((B, H, 4, PITCH, 16), device=dev, dtype=torch.float16)
                    gn_idx = torch.empty((B, H, 4, PITCH), device=dev, dtype=torch.int32)
                    gn_pidx = torch.empty((B, 64, H, 22, 2), device=dev, dtype=torch.int16)
                    part2 = torch.empty((B, 64, H, 6, 2), device=dev, dtype=torch.float32)
                    _hip.call("mx_ln_prelu_bwd_gpool_f16", _hip.ptr(x_in), _hip.ptr(dxhat), _hip.ptr(saved[3 * (l - 1) + 2]),
                              _hip.ptr(stats), _hip.ptr(slope_prev), _hip.ptr(m12), _hip.ptr(scale_n), B, H, n_frames,
                              _hip.ptr(gn_hi), _hip.ptr(gn_lo), _hip.ptr(gn_idx), _hip.ptr(gn_pidx), _hip.ptr(part2),
                              _hip.ptr(ds_part), _hip.ptr(bsum), st)
                    pooled = (gn_hi, gn_lo, gn_idx, gn_pidx, scale_n)
                    gmax_ws, G = None, None
                    del part2, m12, dxhat
                elif pair1:
                    # the first block's gradient as f16x3 pairs, in place: scale from the bound on max|G| (known before the pass)
                    m12 = torch.empty((B, 64, 2), device=dev, dtype=torch.float32)
                    bound_ws = torch.empty(1, device=dev, dtype=torch.int32)
                    pair_scale = torch.empty(2, device=dev, dtype=torch.float32)
                    _hip.call("mx_ln_bwd_finish", _hip.ptr(ln_part), _hip.ptr(stats), _hip.ptr(slope_prev), _hip.ptr(gx_bits),
                              B, 64, H, n_frames, _hip.ptr(m12), _hip.ptr(bound_ws), _hip.ptr(pair_scale), st)
                    _hip.call("mx_ln_prelu_bwd_pair", _hip.ptr(x_in), _hip.ptr(dxhat), _hip.ptr(stats), _hip.ptr(slope_prev),
                              B, 64, H, n_frames, _hip.ptr(ds_part), _hip.ptr(bsum), _hip.ptr(ln_part), _hip.ptr(pair_scale), st)
                    G, gmax_ws = dxhat, None
                    del m12
                else:
                    want_gmax = _use_f16(saved[3 * (l - 1)].size(1), precision) or (l == 1 and 0 in ctx.splits)
                    gmax_ws = zws[4 * l + 2:4 * l + 3] if want_gmax else None
                    _hip.call("mx_ln_prelu_bwd", _hip.ptr(x_in), _hip.ptr(dxhat), _hip.ptr(stats), _hip.ptr(slope_prev),
                              B, 64, H, n_frames, _hip.ptr(ds_part), _hip.ptr(bsum), _hip.ptr(gmax_ws), _hip.ptr(ln_part), st)
                    G = dxhat
                grads[3 * (l - 1) + 2] = _reduce_rows(ds_part, B, 64, gout(3 * (l - 1) + 2))
        if direct is not None:                      # already in place: autograd has nothing to accumulate
            grads = [None] * len(params)
        return (None, None, None, None, *grads)


class Spectral2DCNN(nn.Module):
    def __init__(self,
                 in_ch: int = 1,
                 n_samples: int = 88200,
                 sr: float = 44100,
                 n_fft: int = 1024,
                 hop_len: int = 256,
                 n_mels: int = 256,
                 kernel_size: Tuple[int, int] = (5, 13),
                 out_channels: Optional[List[int]] = None,
                 bin_dilations: Optional[List[int]] = None,
                 temp_dilations: Optional[List[int]] = None,
                 pool_size: Tuple[int, int] = (3, 1),
                 latent_dim: int = 1,
                 freq_mask_amount: float = 0.0,
                 time_mask_amount: float = 0.0,
                 use_ln: bool = True,
                 eps: float = 1e-7) -> None:
        super().__init__()
        if out_channels is None:
            out_channels = [64] * 5
        if bin_dilations is None:
            bin_dilations = [1] * len(out_channels)
        if temp_dilations is None:
            temp_dilations = [2 ** idx for idx in range(len(out_channels))]
        assert len(out_channels) == len(bin_dilations) == len(temp_dilations)
        assert pool_size[1] == 1
        self.sr, self.n_fft, self.hop_len, self.n_mels = sr, n_fft, hop_len, n_mels
        self.kernel_size, self.pool_size, self.latent_dim = tuple(kernel_size), tuple(pool_size), latent_dim
        self.freq_mask_amount, self.time_mask_amount = freq_mask_amount, time_mask_amount
        self.use_ln, self.eps = use_ln, eps
        self.out_channels, self.bin_dilations, self.temp_dilations = list(out_channels), list(bin_dilations), \
            list(temp_dilations)
        self.in_ch = in_ch
        self.n_frames = n_samples // hop_len + 1
        # what the f16x3 / fp32 block kernels are built for (the only configuration the reference ships); everything else -- the
        # class's own defaults included -- runs the general kernels of csrc/cnn_generic.hip (cnn_generic.GenericCNNStack)
        self.generic = (self.kernel_size != (5, 13) or self.pool_size != (2, 1) or not use_ln
                        or any(c != 64 for c in out_channels) or any(d != 1 for d in bin_dilations)
                        or any(d not in (1, 2, 4, 8, 16) for d in temp_dilations) or in_ch not in (1, 2)
                        or self.n_frames > PITCH or n_mels % (2 ** len(out_channels)) != 0 or latent_dim > 4)
        if self.generic:
            n_bins = n_mels
            for _ in out_channels:
                n_bins //= self.pool_size[0]
            if n_bins < 1 or min(self.kernel_size) < 1 or min(self.bin_dilations + self.temp_dilations) < 1 or self.pool_size[0] > 255:
                raise ValueError(f"Spectral2DCNN: {n_mels} mel bins do not survive {len(out_channels)} poolings by "
                                 f"{self.pool_size[0]} (or a kernel size / dilation below 1)")
        self.conv_precision = CONV_PRECISION        # "f16x3" (default) or "f32", see the module header
        self.spectrogram = MelSpectrogramHIP(int(sr), n_fft, hop_len, n_mels)
        self.freq_mask_param = int(freq_mask_amount * n_mels)
        self.time_mask_param = int(time_mask_amount * self.n_frames)
        layers: List[nn.Module] = []
        n_bins, c_in = n_mels, in_ch
        for out_ch, b_dil, t_dil in zip(out_channels, bin_dilations, temp_dilations):
            if use_ln:
                layers.append(nn.LayerNorm([n_bins, self.n_frames], elementwise_affine=False))
            layers.append(nn.Conv2d(c_in, out_ch, self.kernel_size, stride=(1, 1), dilation=(b_dil, t_dil),
                                    padding="same"))
            layers.append(nn.MaxPool2d(kernel_size=self.pool_size))
            layers.append(nn.PReLU(num_parameters=out_ch))
            c_in, n_bins = out_ch, n_bins // self.pool_size[0]
        self.cnn = nn.Sequential(*layers)       # parameter holders; never called
        self.output = nn.Conv1d(out_channels[-1], latent_dim, kernel_size=(1,))

    def _stack_params(self) -> List[T]:
        ps: List[T] = []
        per, first = (4, 1) if self.use_ln else (3, 0)          # module indices as in the reference's nn.Sequential (models.py:184-191)
        for i in range(len(self.out_channels)):
            conv, prelu = self.cnn[per * i + first], self.cnn[per * i + first + 2]
            w = conv.weight
            if i == 0 and self.in_ch == 1 and not self.generic:      # pad the single input channel to the 2-channel kernel
                w = torch.cat([w, torch.zeros_like(w)], dim=1)
            ps += [w, conv.bias, prelu.weight]
        # (the Conv1d's (latent_dim, 64, 1) weight itself, not a 2-D view of it: a view is not a leaf, and its gradient could
        #  not be written in place -- _direct_grad_views; the kernels only need its pointer and first dimension)
        ps += [self.output.weight, self.output.bias]
        return ps

    def draw_masks(self) -> Tuple[int, int, int, int]:
        f0 = f1 = t0 = t1 = 0
        if self.training:
            if self.freq_mask_amount > 0:
                f0, f1 = specaugment_bounds(self.n_mels, self.freq_mask_param)
            if self.time_mask_amount > 0:
                t0, t1 = specaugment_bounds(self.n_frames, self.time_mask_param)
        return f0, f1, t0, t1

    def log_mel(self, x: T, masks: Optional[Sequence[int]] = None) -> T:
        assert x.ndim == 3
        n_frames = x.size(-1) // self.hop_len + 1
        assert n_frames == self.n_frames, "clip length does not match the LayerNorm shape"
        masks = self.draw_masks() if masks is None else masks
        if self.generic:                        # dense (B, in_ch, n_mels, n_frames)
            with torch.no_grad():
                return self.spectrogram.log_mel(x, self.n_frames, self.eps, masks, pitch=self.n_frames)
        if self.in_ch == 1:
            x = torch.cat([x, torch.zeros_like(x)], dim=1)
        with torch.no_grad():
            return self.spectrogram.log_mel(x, self.n_frames, self.eps, masks)

    def forward(self, x: T, masks: Optional[Sequence[int]] = None) -> (T, T):
        logmel = self.log_mel(x, masks)
        if self.generic:
            from .cnn_generic import GenericCNNStack
            cfg = (self.kernel_size, int(self.pool_size[0]), bool(self.use_ln),
                   tuple(zip(self.out_channels, self.bin_dilations, self.temp_dilations)))
            return GenericCNNStack.apply(logmel, cfg, *self._stack_params())
        out, latent = _CNNStack.apply(logmel, self.n_frames, tuple(self.temp_dilations), self.conv_precision,
                                      *self._stack_params())
        return out, latent


class RandomLFO(nn.Module):
    """models.py:19-69: baseline 'extractor' that emits random / perturbed-ground-truth LFOs."""

    def __init__(self,
                 n_samples: int,
                 sr: float,
                 use_shape_gt: bool = False,
                 use_phase_gt: bool = False,
                 use_freq_gt: bool = False,
                 shapes: Optional[List[str]] = None,
                 freq_min: float = 0.5,
                 freq_max: float = 3.0,
                 phase_error: float = 0.0,
                 freq_error: float = 0.0) -> None:
        super().__init__()
        self.n_samples, self.sr = n_samples, sr
        self.use_shape_gt, self.use_phase_gt, self.use_freq_gt = use_shape_gt, use_phase_gt, use_freq_gt
        self.shapes, self.freq_min, self.freq_max = shapes, freq_min, freq_max
        self.phase_error, self.freq_error = phase_error, freq_error

    def forward(self, batch_size: int, fx_params: Optional[Dict[str, T]] = None) -> T:
        from .modulations import make_rand_mod_signal
        shapes_gt = phase_gt = freq_gt = None
        if self.use_shape_gt:
            assert fx_params is not None and "shape" in fx_params
            shapes_gt = fx_params["shape"]
        if self.use_phase_gt:
            assert fx_params is not None and "phase" in fx_params
            phase_gt = fx_params["phase"]
        if self.use_freq_gt:
            assert fx_params is not None and "rate_hz" in fx_params
            freq_gt = fx_params["rate_hz"]
        return make_rand_mod_signal(batch_size, self.n_samples, self.sr, self.freq_min, self.freq_max, shapes_gt,
                                    self.shapes, phase_gt, self.phase_error, freq_gt, self.freq_error).unsqueeze(1)


class SpectrogramHIP(nn.Module):
    """State-dict compatible stand-in for ``torchaudio.transforms.Spectrogram(n_fft, hop_length=hop, normalized=False)``
    (key ``window``; power 2, centre + reflect padding) of SpectralTCN / SpectralDSTCN (models.py:99,252): the log-mel
    kernel with an identity filter bank -- band m = bin m, weight 1.0 -- returns log(clip(|STFT|^2, eps)) exactly."""

    def __init__(self, n_fft: int, hop_length: int) -> None:
        super().__init__()
        if n_fft not in (512, 1024, 2048):
            raise NotImplementedError("mx_logmel_fwd is built for n_fft in {512, 1024, 2048}")
        self.n_fft, self.hop_length, self.n_bins = n_fft, hop_length, n_fft // 2 + 1
        self.register_buffer("window", torch.hann_window(n_fft))
        k = torch.arange(n_fft, dtype=torch.float64) * (-2.0 * math.pi / n_fft)
        self.register_buffer("twiddle", torch.stack([torch.cos(k), torch.sin(k)], dim=1).float(), persistent=False)
        self.register_buffer("eye", torch.eye(self.n_bins), persistent=False)
        self.register_buffer("band_lo", torch.arange(self.n_bins, dtype=torch.int32), persistent=False)
        self.register_buffer("band_hi", torch.arange(1, self.n_bins + 1, dtype=torch.int32), persistent=False)

    def log_power(self, x: T, n_frames: int, eps: float) -> T:
        """x (B, 1, N) -> (B, n_fft/2 + 1, 352) planes = log(clip(power, eps)), columns >= n_frames zero"""
        assert x.ndim == 3 and x.size(1) == 1
        B, _, N = x.shape
        xc = x.contiguous().float()
        out = torch.empty((B, self.n_bins, PITCH), device=x.device, dtype=torch.float32)
        _hip.call("mx_logmel_fwd", _hip.ptr(xc), B, N, _hip.ptr(self.window), _hip.ptr(self.twiddle), _hip.ptr(self.eye),
                  _hip.ptr(self.band_lo), _hip.ptr(self.band_hi), self.n_fft, self.hop_length, self.n_bins, n_frames, PITCH,
                  float(eps), 0, 0, 0, 0, _hip.ptr(out), _hip.stream())
        return out


class SpectralTCN(nn.Module):
    """models.py:72-125: log power spectrogram (513 bins) -> 5-block dilated TCN over time (LayerNorm, 13 taps, PReLU,
    1x1 residual) -> Conv1d(96 -> latent_dim, 1) -> sigmoid; (B, 1, N) -> (B, latent_dim, frames).  The front end and the
    TCN stack run in HIP kernels (``tcn.py``); the 1x1 head + sigmoid is ``mx_binmean_head_*`` at one bin."""

    def __init__(self, n_samples: int = 88200, n_fft: int = 1024, hop_len: int = 256, kernel_size: int = 13,
                 out_channels: Optional[List[int]] = None, dilations: Optional[List[int]] = None, latent_dim: int = 1,
                 use_ln: bool = True, use_res: bool = True, eps: float = 1e-7) -> None:
        super().__init__()
        from .tcn import TCN
        self.n_fft, self.hop_len, self.kernel_size, self.latent_dim = n_fft, hop_len, kernel_size, latent_dim
        self.use_ln, self.use_res, self.eps = use_ln, use_res, eps
        if out_channels is None:
            out_channels = [96] * 5
        self.out_channels = out_channels
        if dilations is None:
            dilations = [2 ** idx for idx in range(len(out_channels))]
        self.dilations = dilations
        self.spectrogram = SpectrogramHIP(n_fft, hop_len)
        self.n_frames = n_samples // hop_len + 1
        self.tcn = TCN(out_channels, dilations, n_fft // 2 + 1, kernel_size, padding=None, use_ln=use_ln,
                       temporal_dims=[self.n_frames] * len(out_channels), use_res=use_res, is_causal=False)
        self.receptive_field = self.tcn.calc_receptive_field()
        log.info(f"Receptive field = {self.receptive_field} samples")
        self.output = nn.Conv1d(out_channels[-1], self.latent_dim, kernel_size=(1,))

    def features(self, x: T) -> T:
        assert x.ndim == 3
        n_frames = x.size(-1) // self.hop_len + 1
        with torch.no_grad():
            spec = self.spectrogram.log_power(x, n_frames, self.eps)
        y, t_out = self.tcn.forward_planes(spec, n_frames)
        return y[:, :, :t_out]

    def forward(self, x: T) -> T:
        from .cnn_generic import BinMeanHead
        f = self.features(x)                                        # (B, C, T'): Conv1d(C, L, 1) + sigmoid = the bin-mean head at one bin
        y, _ = BinMeanHead.apply(f.unsqueeze(2), self.output.weight, self.output.bias)
        return y


class SpectralDSTCN(nn.Module):
    """models.py:218-289: the strided (down-sampling) variant: TCN with stride 2 per block -> mean over time ->
    Linear(96, 48) -> PReLU -> Linear(48, latent_dim) -> sigmoid; (B, 1, N) -> (B, latent_dim)."""

    def __init__(self, n_samples: int = 88200, n_fft: int = 1024, hop_len: int = 256, kernel_size: int = 13,
                 out_channels: Optional[List[int]] = None, dilations: Optional[List[int]] = None,
                 strides: Optional[List[int]] = None, n_fc_units: int = 48, latent_dim: int = 2, use_ln: bool = True,
                 use_res: bool = True, eps: float = 1e-7) -> None:
        super().__init__()
        from .tcn import TCN
        self.n_fft, self.hop_len, self.kernel_size, self.n_fc_units, self.latent_dim = n_fft, hop_len, kernel_size, n_fc_units, latent_dim
        self.use_ln, self.use_res, self.eps = use_ln, use_res, eps
        if out_channels is None:
            out_channels = [96] * 5
        self.out_channels = out_channels
        if dilations is None:
            dilations = [2 ** idx for idx in range(len(out_channels))]
        self.dilations = dilations
        if strides is None:
            strides = [2] * len(out_channels)
        self.strides = strides
        self.spectrogram = SpectrogramHIP(n_fft, hop_len)
        self.n_frames = n_samples // hop_len + 1
        temporal_dims, cur = [self.n_frames], self.n_frames
        for stride in strides[:-1]:
            cur = math.ceil(cur / stride)
            temporal_dims.append(cur)
        self.tcn = TCN(out_channels, dilations, n_fft // 2 + 1, kernel_size, strides, padding=None, use_ln=use_ln,
                       temporal_dims=temporal_dims, use_res=use_res, is_causal=False)
        self.fc = nn.Linear(out_channels[-1], self.n_fc_units)
        self.fc_act = nn.PReLU(self.n_fc_units)
        self.output = nn.Linear(self.n_fc_units, self.latent_dim)

    def forward(self, x: T) -> T:
        assert x.ndim == 3
        n_frames = x.size(-1) // self.hop_len + 1
        with torch.no_grad():
            spec = self.spectrogram.log_power(x, n_frames, self.eps)
        y, t_out = self.tcn.forward_planes(spec, n_frames)
        from .cnn_generic import BinMeanHead, LinearPReLU, time_mean
        f = time_mean(y[:, :, :t_out])                              # models.py:283: mean over the remaining frames
        hid = LinearPReLU.apply(f, self.fc.weight, self.fc.bias, self.fc_act.weight)
        out, _ = BinMeanHead.apply(hid.view(hid.size(0), -1, 1, 1), self.output.weight, self.output.bias)   # Linear + sigmoid
        return out.view(out.size(0), -1)


class HiddenStateModel(nn.Module):
    """models.py:292-308."""

    def __init__(self) -> None:
        super().__init__()
        self.hidden: Tuple[T, T] = (torch.zeros((1,)), torch.zeros((1,)))
        self.is_hidden_init = False

    def update_hidden(self, hidden: Tuple[T, T]) -> None:
        self.hidden = hidden
        self.is_hidden_init = True

    def detach_hidden(self) -> None:
        if self.is_hidden_init:
            self.hidden = tuple((h.detach().clone() for h in self.hidden))

    def clear_hidden(self) -> None:
        self.is_hidden_init = False


LSTM_NPARAM = 17473        # 256*2 + 256*64 + 256 + 256 + 64 + 1 (csrc/lstm.hip)


def _rows(t: T) -> Tuple[int, int]:
    """(device pointer, row stride) of a (B,1,T)/(B,T) fp32 view whose rows are contiguous."""
    if not t.is_cuda:
        raise _hip.HipLibraryError("mod_extraction_amd ops need tensors on a HIP device (no CPU fallback)")
    if t.ndim == 3:
        assert t.size(1) == 1
        t = t[:, 0, :]
    assert t.dtype == torch.float32 and t.ndim == 2 and t.stride(1) == 1
    return t.data_ptr(), t.stride(0)


class LSTMEffectModel(HiddenStateModel):
    """models.py:311-339.  ``self.lstm`` / ``self.fc`` hold the parameters under the reference's
    state-dict keys (the 7 shipped ``models/lstm_64__*.pt`` files load with strict=True); the recurrence
    runs in ``mx_lstm_fwd`` (one workgroup per clip, weights in registers, state in LDS)."""

    def __init__(self, in_ch: int = 1, out_ch: int = 1, n_hidden: int = 64, latent_dim: int = 1) -> None:
        super().__init__()
        # the fused kernels of csrc/lstm.hip are the shipped LSTM-64 with one audio and one LFO channel; any other size runs
        # the general recurrence of csrc/lstm_generic.hip as an autograd node (lstm_generic.GenericLSTM)
        self.generic = (in_ch, out_ch, n_hidden, latent_dim) != (1, 1, 64, 1)
        if self.generic and not (out_ch == in_ch or out_ch == 1 or in_ch == 1):
            raise ValueError("LSTMEffectModel: fc output (out_ch) and x (in_ch) do not broadcast (models.py:338)")
        self.in_ch, self.out_ch, self.n_hidden, self.latent_dim = in_ch, out_ch, n_hidden, latent_dim
        self.lstm = nn.LSTM(in_ch + latent_dim, n_hidden, batch_first=True)     # parameter holder
        self.fc = nn.Linear(n_hidden, out_ch)                                    # parameter holder

    def _params(self) -> List[T]:
        return [self.lstm.weight_ih_l0, self.lstm.weight_hh_l0, self.lstm.bias_ih_l0, self.lstm.bias_hh_l0,
                self.fc.weight, self.fc.bias]

    def _state(self, B: int, device) -> Tuple[T, T]:
        if self.is_hidden_init:
            h, c = self.hidden
            return h.reshape(B, self.n_hidden), c.reshape(B, self.n_hidden)
        return (torch.zeros((B, self.n_hidden), device=device, dtype=torch.float32),
                torch.zeros((B, self.n_hidden), device=device, dtype=torch.float32))

    def detach_hidden(self) -> None:
        """models.py:303-305 clones the detached state; the state tensors here never carry a graph and the kernels never
        write to a state they were given as input (``mx_lstm_fwd`` writes the new state to fresh buffers), so there is
        nothing to copy."""

    def run_chunk(self, x: T, latent: T, stash: Optional[T] = None) -> Tuple[T, T, T]:
        """Forward one chunk without autograd.  Returns (y (B,1,T), h_start, c_start); updates the hidden
        state.  ``stash`` (B,T,384) receives the per-step activations when a BPTT step follows."""
        assert x.ndim == 3 and latent.shape == (x.size(0), self.latent_dim, x.size(-1))
        B, _, Tn = x.shape
        h0, c0 = self._state(B, x.device)
        h1 = torch.empty((B, 64), device=x.device, dtype=torch.float32)
        c1 = torch.empty((B, 64), device=x.device, dtype=torch.float32)
        y = torch.empty((B, 1, Tn), device=x.device, dtype=torch.float32)
        xp, xs = _rows(x)
        lp, ls = _rows(latent)
        yp, ys = _rows(y)
        w = [p.detach().contiguous() for p in self._params()]
        _hip.call("mx_lstm_fwd", xp, xs, lp, ls, _hip.ptr(w[0]), _hip.ptr(w[1]), _hip.ptr(w[2]), _hip.ptr(w[3]),
                  _hip.ptr(w[4]), _hip.ptr(w[5]), _hip.ptr(h0.contiguous()), _hip.ptr(c0.contiguous()), _hip.ptr(h1), _hip.ptr(c1),
                  yp, ys, _hip.ptr(stash), B, Tn, _hip.stream())
        self.update_hidden((h1.view(1, B, 64), c1.view(1, B, 64)))
        return y, h0, c0

    def bptt_l1_chunk(self, x: T, latent: T, y: T, wet: T, stash: T, h0: T, c0: T, loss_scale: float,
                      grad_out: Optional[T]) -> Optional[T]:
        """BPTT of one chunk with the L1 loss fused; the summed parameter gradient (17473,) in
        state-dict order is written to ``grad_out``.  ``grad_out=None``: the per-clip gradient rows (B, 17473) are returned
        unsummed (``FlatAdamW.step_from_rows`` sums them and steps in one launch)."""
        B, _, Tn = x.shape
        assert grad_out is None or (grad_out.numel() == LSTM_NPARAM and grad_out.is_contiguous())     # every element is overwritten below
        part = torch.empty((B, LSTM_NPARAM), device=x.device, dtype=torch.float32)
        xp, xs = _rows(x)
        lp, ls = _rows(latent)
        yp, ys = _rows(y)
        wp, ws = _rows(wet)
        _hip.call("mx_lstm_bwd_l1", xp, xs, lp, ls, yp, ys, wp, ws, _hip.ptr(stash),
                  _hip.ptr(self.lstm.weight_hh_l0.detach().contiguous()), _hip.ptr(self.fc.weight.detach().contiguous()),
                  _hip.ptr(h0.contiguous()), _hip.ptr(c0.contiguous()), float(loss_scale), _hip.ptr(part), B, Tn, _hip.stream())
        if grad_out is None:
            return part
        _hip.call("mx_reduce_rows", _hip.ptr(part), B, LSTM_NPARAM, 0, _hip.ptr(grad_out), _hip.stream())
        return None

    def bptt_chunk(self, x: T, latent: T, y: T, dy: T, stash: T, h0: T, c0: T, grad_out: T) -> None:
        """BPTT of one chunk for ANY loss: ``dy`` (B,1,T) or (B,T) = d loss / d y (``effect_losses.effect_loss_grad``);
        the summed parameter gradient (17473,) in state-dict order is written to ``grad_out``."""
        B, _, Tn = x.shape
        assert grad_out.numel() == LSTM_NPARAM and grad_out.is_contiguous()
        dy = dy.view(B, Tn)
        assert dy.stride(1) == 1
        part = torch.empty((B, LSTM_NPARAM), device=x.device, dtype=torch.float32)
        xp, xs = _rows(x)
        lp, ls = _rows(latent)
        yp, ys = _rows(y)
        _hip.call("mx_lstm_bwd", xp, xs, lp, ls, yp, ys, dy.data_ptr(), dy.stride(0), _hip.ptr(stash),
                  _hip.ptr(self.lstm.weight_hh_l0.detach().contiguous()), _hip.ptr(self.fc.weight.detach().contiguous()),
                  _hip.ptr(h0.contiguous()), _hip.ptr(c0.contiguous()), _hip.ptr(part), B, Tn, _hip.stream())
        _hip.call("mx_reduce_rows", _hip.ptr(part), B, LSTM_NPARAM, 0, _hip.ptr(grad_out), _hip.stream())

    def bptt_chunk_dlfo(self, x: T, latent: T, y: T, stash: T, h0: T, c0: T, grad_out: T, wet: Optional[T] = None,
                        loss_scale: float = 0.0, dy: Optional[T] = None) -> T:
        """``bptt_l1_chunk`` (``wet`` + ``loss_scale``) or ``bptt_chunk`` (``dy``) that also returns d loss / d latent (B,1,T):
        the gradient an UNFROZEN LFO model receives through the LFO it produced (lightning.py:258,361).  The kernel leaves the
        gate gradients (B,T,256); ``mx_lstm_dlfo`` contracts them with the LFO column of ``weight_ih_l0``."""
        B, _, Tn = x.shape
        assert grad_out.numel() == LSTM_NPARAM and grad_out.is_contiguous()
        assert (wet is None) != (dy is None)
        part = torch.empty((B, LSTM_NPARAM), device=x.device, dtype=torch.float32)
        dgate = torch.empty((B, Tn, 256), device=x.device, dtype=torch.float32)
        dlat = torch.empty((B, 1, Tn), device=x.device, dtype=torch.float32)
        xp, xs = _rows(x)
        lp, ls = _rows(latent)
        yp, ys = _rows(y)
        wp, ws = _rows(wet) if wet is not None else (None, 0)
        if dy is not None:
            dy = dy.view(B, Tn)
            assert dy.stride(1) == 1
        _hip.call("mx_lstm_bwd_dgate", xp, xs, lp, ls, yp, ys, wp, ws, None if dy is None else dy.data_ptr(),
                  0 if dy is None else dy.stride(0), _hip.ptr(stash), _hip.ptr(self.lstm.weight_hh_l0.detach().contiguous()),
                  _hip.ptr(self.fc.weight.detach().contiguous()), _hip.ptr(h0.contiguous()), _hip.ptr(c0.contiguous()),
                  float(loss_scale), _hip.ptr(part), _hip.ptr(dgate), B, Tn, _hip.stream())
        _hip.call("mx_reduce_rows", _hip.ptr(part), B, LSTM_NPARAM, 0, _hip.ptr(grad_out), _hip.stream())
        _hip.call("mx_lstm_dlfo", _hip.ptr(dgate), _hip.ptr(self.lstm.weight_ih_l0.detach().contiguous()), B, Tn,
                  _hip.ptr(dlat), Tn, _hip.stream())
        return dlat

    def forward(self, x: T, latent: T) -> T:
        """Inference / validation forward (no autograd graph; training goes through the fused TBPTT
        step of ``lightning.TBPTTLFOEffectModeling``).  A model of another size (``self.generic``) is an ordinary autograd node:
        gradients reach its parameters and ``latent``; the carried state is a constant (detached, as lightning.py:353,383 do)."""
        if self.generic:
            from .lstm_generic import GenericLSTM
            assert x.ndim == 3 and latent.shape == (x.size(0), self.latent_dim, x.size(-1)) and x.size(1) == self.in_ch
            B = x.size(0)
            h0, c0 = self._state(B, x.device)
            y, h1, c1 = GenericLSTM.apply(x, latent, h0, c0, *self._params())
            self.update_hidden((h1.view(1, B, self.n_hidden), c1.view(1, B, self.n_hidden)))
            return y
        with torch.no_grad():
            y, _, _ = self.run_chunk(x.contiguous().float(), latent.contiguous().float())
        return y
