"""``Spectral2DCNN`` (mod_extraction/models.py:127-215) outside the family the f16x3 kernels are built for: any kernel
size, channel list, bin / frame dilations, ``MaxPool2d((p, 1))``, with or without LayerNorm, any ``in_ch`` / ``latent_dim`` /
frame count -- e.g. the class's own defaults (``pool_size=(3, 1)``, five blocks).  One autograd node over the whole stack,
every step a kernel of ``csrc/cnn_generic.hip`` or the fp32 matrix-core GEMM of ``csrc/tcn.hip``:

    [LayerNorm([bins, frames])] -> im2col -> GEMM (+ bias) -> MaxPool2d((p, 1)) + PReLU        per block (models.py:183-191)
    mean over bins -> Conv1d(C, L, 1) -> sigmoid                                               (models.py:209-215)

Dense NCHW fp32 tensors, exact fp32 products.  The im2col matrix of a block is built for a chunk of clips at a time
(``COL_BYTES`` per chunk) and rebuilt in the backward pass for the weight gradient.
"""
import os
from typing import List, Tuple

import torch
from torch import Tensor as T

from . import _hip
from .tcn import _sgemm

LN_EPS = 1e-5                                                       # torch.nn.LayerNorm default
COL_BYTES = int(os.environ.get("MODEX_GENERIC_COL_MB", "1024")) << 20


def same_padding(k: int, d: int) -> int:
    """Leading padding of Conv2d(padding="same") along one axis (aten: total = d (k - 1), total // 2 before, the rest after)."""
    return (d * (k - 1)) // 2


def _empty(shape, dev, dtype=torch.float32) -> T:
    return torch.empty(shape, device=dev, dtype=dtype)


def _chunks(B: int, per_clip_bytes: int) -> List[Tuple[int, int]]:
    nb = max(1, min(B, COL_BYTES // max(1, per_clip_bytes)))
    return [(b0, min(nb, B - b0)) for b0 in range(0, B, nb)]


class GenericCNNStack(torch.autograd.Function):
    """logmel (B, Cin, H, W) dense -> (sigmoid output (B, L, W), latent (B, C_last, W)).
    ``cfg`` = (kernel_size, pool, use_ln, ((cout, bin_dilation, temp_dilation), ...));
    ``params`` = [w1, b1, a1, ..., wn, bn, an, wout, bout] (``nn.Conv2d`` / ``nn.PReLU`` / ``nn.Conv1d`` tensors)."""

    @staticmethod
    def forward(ctx, logmel: T, cfg, *params: T):
        (kh, kw), pool, use_ln, blocks = cfg
        dev, st = logmel.device, _hip.stream()
        cur = logmel.contiguous().float()
        B, cin, H, W = cur.shape
        saved: List[T] = []
        geoms = []
        for l, (cout, bd, td) in enumerate(blocks):
            w, b, a = (params[3 * l + i].detach().contiguous().float() for i in range(3))
            assert w.shape == (cout, cin, kh, kw)
            if use_ln:
                xhat, stats = _empty((B, cin, H, W), dev), _empty((B * cin, 2), dev)
                _hip.call("mx_rowln_fwd", _hip.ptr(cur), B * cin, H * W, LN_EPS, _hip.ptr(xhat), _hip.ptr(stats), st)
            else:
                xhat, stats = cur, None
            K, HW = cin * kh * kw, H * W
            pt, pl = same_padding(kh, bd), same_padding(kw, td)
            z = _empty((B, cout, H, W), dev)
            for b0, nb in _chunks(B, K * HW * 4):
                col = _empty((K, nb * HW), dev)
                _hip.call("mx_im2col2d", _hip.ptr(xhat[b0]), nb, cin, H, W, kh, kw, bd, td, pt, pl, _hip.ptr(col), st)
                _sgemm(_hip.ptr(w), K, 1, 0, _hip.ptr(col), nb * HW, 1, HW, _hip.ptr(z[b0]), HW, 1, cout * HW, cout, HW, K, nb)
                del col
            Hp = H // pool
            assert Hp >= 1, "MaxPool2d: fewer bins than the pooling window"
            v, out = _empty((B, cout, Hp, W), dev), _empty((B, cout, Hp, W), dev)
            amax = _empty((B, cout, Hp, W), dev, torch.uint8)
            _hip.call("mx_pool_prelu_fwd", _hip.ptr(z), _hip.ptr(b), B * cout, cout, H, W, pool, _hip.ptr(a), _hip.ptr(v), _hip.ptr(out),
                      _hip.ptr(amax), st)
            del z
            saved += [xhat, stats if stats is not None else _empty((0,), dev), v, amax]
            geoms.append((cin, cout, H, bd, td))
            cur, cin, H = out, cout, Hp
        wout, bout = params[-2].detach().contiguous().float(), params[-1].detach().contiguous().float()
        L = wout.size(0)
        latent, y = _empty((B, cin, W), dev), _empty((B, L, W), dev)
        _hip.call("mx_binmean_head_fwd", _hip.ptr(cur), B, cin, H, W, _hip.ptr(wout), _hip.ptr(bout), L, _hip.ptr(latent), _hip.ptr(y),
                  st)
        ctx.cfg, ctx.geoms, ctx.head = cfg, geoms, (B, cin, H, W, L)
        ctx.save_for_backward(*saved, latent, y, *[p.detach() for p in params])
        return y, latent

    @staticmethod
    def backward(ctx, d_out, d_latent):
        (kh, kw), pool, use_ln, blocks = ctx.cfg
        n = len(blocks)
        saved = ctx.saved_tensors
        latent, y = saved[4 * n], saved[4 * n + 1]
        params = saved[4 * n + 2:]
        B, c_last, H_last, W, L = ctx.head
        dev, st = y.device, _hip.stream()
        wout = params[-2].contiguous().float()
        grads: List[T] = [None] * len(params)
        ds, g = _empty((B, L, W), dev), _empty((B, c_last, H_last, W), dev)
        _hip.call("mx_binmean_head_bwd", _hip.ptr(d_out.contiguous().float()) if d_out is not None else None,
                  _hip.ptr(d_latent.contiguous().float()) if d_latent is not None else None, _hip.ptr(y), _hip.ptr(wout), B, c_last,
                  H_last, W, L, _hip.ptr(ds), _hip.ptr(g), st)
        # Conv1d(C, L, 1): dW[l][c] = sum over clips and frames of ds[b][l][w] latent[b][c][w]; db[l] = sum of ds
        dwo = _empty((L, c_last), dev)
        _sgemm(_hip.ptr(ds), W, 1, L * W, _hip.ptr(latent), 1, W, c_last * W, _hip.ptr(dwo), c_last, 1, 0, L, c_last, W, B, per_group=B)
        rs = _empty((B * L,), dev)
        _hip.call("mx_row_sums", _hip.ptr(ds), B * L, W, _hip.ptr(rs), st)
        dbo = _empty((L,), dev)
        _hip.call("mx_reduce_rows", _hip.ptr(rs), B, L, 0, _hip.ptr(dbo), st)
        grads[-2], grads[-1] = dwo.view(params[-2].shape), dbo
        need_x = ctx.needs_input_grad[0]
        for l in range(n - 1, -1, -1):
            cin, cout, H, bd, td = ctx.geoms[l]
            xhat, stats, v, amax = saved[4 * l:4 * l + 4]
            w, a = params[3 * l].contiguous().float(), params[3 * l + 2].contiguous().float()
            dz, part = _empty((B, cout, H, W), dev), _empty((B * cout, 2), dev)
            _hip.call("mx_pool_prelu_bwd", _hip.ptr(g), _hip.ptr(v), _hip.ptr(amax), B * cout, cout, H, W, pool, _hip.ptr(a), _hip.ptr(dz),
                      _hip.ptr(part), st)
            red = _empty((cout, 2), dev)
            _hip.call("mx_reduce_rows", _hip.ptr(part), B, cout * 2, 0, _hip.ptr(red), st)
            grads[3 * l + 1], grads[3 * l + 2] = red[:, 0].contiguous(), red[:, 1].contiguous().view(params[3 * l + 2].shape)
            K, HW = cin * kh * kw, H * W
            pt, pl = same_padding(kh, bd), same_padding(kw, td)
            dw = _empty((cout, K), dev)
            want_dx = l > 0 or need_x
            dxh = _empty((B, cin, H, W), dev) if want_dx else None
            for ci, (b0, nb) in enumerate(_chunks(B, K * HW * 4)):
                col = _empty((K, nb * HW), dev)
                _hip.call("mx_im2col2d", _hip.ptr(xhat[b0]), nb, cin, H, W, kh, kw, bd, td, pt, pl, _hip.ptr(col), st)
                # dW[co][k] = sum over the chunk's clips and positions of dz[b][co][p] col[k][b HW + p]: one partial per clip
                pw = _empty((nb, cout, K), dev)
                _sgemm(_hip.ptr(dz[b0]), HW, 1, cout * HW, _hip.ptr(col), 1, nb * HW, HW, _hip.ptr(pw), K, 1, cout * K, cout, K, HW, nb)
                _hip.call("mx_reduce_rows", _hip.ptr(pw), nb, cout * K, 1 if ci else 0, _hip.ptr(dw), st)
                if want_dx:
                    # dcol[k][b HW + p] = sum over co of w[co][k] dz[b][co][p] (into the same buffer), then the transposed gather
                    _sgemm(_hip.ptr(w), 1, K, 0, _hip.ptr(dz[b0]), HW, 1, cout * HW, _hip.ptr(col), nb * HW, 1, HW, K, HW, cout, nb)
                    _hip.call("mx_col2im2d", _hip.ptr(col), nb, cin, H, W, kh, kw, bd, td, pt, pl, _hip.ptr(dxh[b0]), st)
                del col, pw
            grads[3 * l] = dw.view(params[3 * l].shape)
            del dz
            if want_dx:
                if use_ln:
                    g = _empty((B, cin, H, W), dev)
                    _hip.call("mx_rowln_bwd", _hip.ptr(dxh), _hip.ptr(xhat), _hip.ptr(stats), B * cin, HW, _hip.ptr(g), st)
                else:
                    g = dxh
        return (g if need_x else None, None, *grads)


class BinMeanHead(torch.autograd.Function):
    """x (B, C, H, W) -> (sigmoid(Conv1d(C, L, 1)(mean over H)) (B, L, W), the mean (B, C, W)): models.py:209-215 on its own --
    also the SpectralTCN head (H = 1: models.py:121-124) and, with H = W = 1, a Linear + sigmoid on (B, C) vectors."""

    @staticmethod
    def forward(ctx, x: T, wout: T, bout: T):
        x = x.contiguous().float()
        B, C, H, W = x.shape
        w, b = wout.detach().contiguous().float(), bout.detach().contiguous().float()
        L = w.size(0)
        latent, y = _empty((B, C, W), x.device), _empty((B, L, W), x.device)
        _hip.call("mx_binmean_head_fwd", _hip.ptr(x), B, C, H, W, _hip.ptr(w), _hip.ptr(b), L, _hip.ptr(latent), _hip.ptr(y), _hip.stream())
        ctx.dims = (B, C, H, W, L)
        ctx.save_for_backward(latent, y, w)
        ctx.w_shape = wout.shape
        return y, latent

    @staticmethod
    def backward(ctx, d_out, d_latent):
        latent, y, w = ctx.saved_tensors
        B, C, H, W, L = ctx.dims
        dev, st = y.device, _hip.stream()
        ds, dx = _empty((B, L, W), dev), _empty((B, C, H, W), dev)
        _hip.call("mx_binmean_head_bwd", _hip.ptr(d_out.contiguous().float()) if d_out is not None else None,
                  _hip.ptr(d_latent.contiguous().float()) if d_latent is not None else None, _hip.ptr(y), _hip.ptr(w), B, C, H, W, L,
                  _hip.ptr(ds), _hip.ptr(dx), st)
        dwo = _empty((L, C), dev)
        _sgemm(_hip.ptr(ds), W, 1, L * W, _hip.ptr(latent), 1, W, C * W, _hip.ptr(dwo), C, 1, 0, L, C, W, B, per_group=B)
        rs, dbo = _empty((B * L,), dev), _empty((L,), dev)
        _hip.call("mx_row_sums", _hip.ptr(ds), B * L, W, _hip.ptr(rs), st)
        _hip.call("mx_reduce_rows", _hip.ptr(rs), B, L, 0, _hip.ptr(dbo), st)
        return dx, dwo.view(ctx.w_shape), dbo


class LinearPReLU(torch.autograd.Function):
    """(B, Cin) -> PReLU(Linear(Cin, Cout)) (B, Cout): the hidden layer of the SpectralDSTCN head (models.py:284-287)."""

    @staticmethod
    def forward(ctx, x: T, w: T, b: T, slope: T):
        x = x.contiguous().float()
        wc, bc, sc = (t.detach().contiguous().float() for t in (w, b, slope))
        B, cin, cout, st = x.size(0), x.size(1), wc.size(0), _hip.stream()
        a = bc.view(1, cout).expand(B, cout).contiguous()                        # the products accumulate onto the bias
        _sgemm(_hip.ptr(x), cin, 1, 0, _hip.ptr(wc), 1, cin, 0, _hip.ptr(a), cout, 1, 0, B, cout, cin, 1, accumulate=1)
        y = _empty((B, cout), x.device)
        _hip.call("mx_prelu_res_fwd", _hip.ptr(a), _hip.ptr(sc), None, B, cout, 1, _hip.ptr(y), st)
        ctx.save_for_backward(x, a, wc, sc)
        return y

    @staticmethod
    def backward(ctx, dy: T):
        x, a, w, slope = ctx.saved_tensors
        B, cin, cout, dev, st = x.size(0), x.size(1), w.size(0), x.device, _hip.stream()
        da, part = _empty((B, cout), dev), _empty((B * cout,), dev)
        _hip.call("mx_prelu_res_bwd", _hip.ptr(dy.contiguous().float()), _hip.ptr(a), _hip.ptr(slope), B, cout, 1, _hip.ptr(da), _hip.ptr(part), st)
        d_slope, d_b = _empty((cout,), dev), _empty((cout,), dev)
        _hip.call("mx_reduce_rows", _hip.ptr(part), B, cout, 0, _hip.ptr(d_slope), st)
        _hip.call("mx_reduce_rows", _hip.ptr(da), B, cout, 0, _hip.ptr(d_b), st)
        d_w, d_x = _empty((cout, cin), dev), _empty((B, cin), dev)
        _sgemm(_hip.ptr(da), 1, cout, 0, _hip.ptr(x), cin, 1, 0, _hip.ptr(d_w), cin, 1, 0, cout, cin, B, 1)
        _sgemm(_hip.ptr(da), cout, 1, 0, _hip.ptr(w), cin, 1, 0, _hip.ptr(d_x), cin, 1, 0, B, cin, cout, 1)
        return d_x, d_w, d_b, d_slope


def time_mean(x: T) -> T:
    """(B, C, T) dense -> (B, C): the mean over frames through the bin-mean kernel (frames in the bins' place, one column)."""
    B, C, Tn = x.shape
    zero_w = torch.zeros((1, C), device=x.device, dtype=torch.float32)
    _, latent = BinMeanHead.apply(x.contiguous().view(B, C, Tn, 1), zero_w, zero_w[0, :1])
    return latent.view(B, C)
