"""Host mirror of mod_extraction/tcn.py: ``TCN`` / ``TCNBlock`` with the reference's constructor arguments, module tree
and state-dict keys (``blocks.<i>.conv.weight``, ``.conv.bias``, ``.act.weight``, ``.res.weight``); the arithmetic runs in
``csrc/tcn.hip`` (LayerNorm statistics -> im2col gather -> fp32 GEMM on the matrix cores -> bias / PReLU / residual, and
the matching backward) as ONE autograd node over the whole stack.

Two routes.  What ``SpectralTCN`` / ``SpectralDSTCN`` (models.py:72-125,218-289) use -- non-causal blocks with automatic
padding (``padding=None``), optional LayerNorm / PReLU / residual, any dilation and stride, at most 352 frames -- runs as ONE
autograd node over 352-column planes (``_TCNStack``).  Everything else of the reference's class -- explicit padding with the
causal / centre crop of the residual (tcn.py:14-29,188-191; ``is_causal`` is the class default), the cached streaming convolution
(``is_cached``: ``Conv1dCached`` / ``PaddingCached``, tcn.py:31-79), FiLM conditioning with or without its BatchNorm1d
(tcn.py:82-103), even kernel sizes, longer clips -- runs block by block on dense (B, C, T) tensors (``_GeneralBlockFn``:
``csrc/tcn_general.hip`` + ``mx_im2col2d`` + ``mx_sgemm_f32`` + ``mx_rowln_*``).
"""
import logging
import os
from typing import List, Optional, Tuple

import torch
from torch import Tensor, nn

from . import _hip

log = logging.getLogger(__name__)
log.setLevel(level=os.environ.get("LOGLEVEL", "INFO"))

PITCH = 352


def center_crop(x: Tensor, length: int) -> Tensor:
    """tcn.py:14-20."""
    if x.size(-1) != length:
        assert x.size(-1) > length
        start = (x.size(-1) - length) // 2
        x = x[..., start:start + length]
    return x


def causal_crop(x: Tensor, length: int) -> Tensor:
    """tcn.py:23-29."""
    if x.size(-1) != length:
        assert x.size(-1) > length
        stop = x.size(-1) - 1
        x = x[..., stop - length:stop]
    return x


def _sgemm(a, a_rs, a_cs, a_bs, b, b_rs, b_cs, b_bs, c, c_rs, c_cs, c_bs, M, N, K, n_batch, per_group=1, accumulate=0):
    _hip.call("mx_sgemm_f32", a, a_rs, a_cs, a_bs, b, b_rs, b_cs, b_bs, c, c_rs, c_cs, c_bs, M, N, K, n_batch, per_group,
              accumulate, _hip.stream())


def _reduce_rows(part: Tensor, R: int, C: int) -> Tensor:
    out = torch.empty(C, device=part.device, dtype=torch.float32)
    _hip.call("mx_reduce_rows", _hip.ptr(part), R, C, 0, _hip.ptr(out), _hip.stream())
    return out


class _TCNStack(torch.autograd.Function):
    """x (B, C0, 352) planes (T0 valid frames) -> (B, C_last, 352).  ``cfg``: one tuple per block
    (cin, cout, ksz, dilation, stride, use_ln, use_act, use_res, t_in, t_out, eps); ``params``: per block conv.weight,
    conv.bias, then act.weight if use_act, then res.weight if use_res."""

    @staticmethod
    def forward(ctx, x: Tensor, cfg: Tuple, need_input_grad: bool, *params: Tensor):
        B = x.size(0)
        dev = x.device
        saved, pi = [], 0
        cur = x.contiguous()
        for (cin, cout, ksz, dil, stride, use_ln, use_act, use_res, t_in, t_out, eps) in cfg:
            w, bias = params[pi], params[pi + 1]
            pi += 2
            slope = res_w = None
            if use_act:
                slope, pi = params[pi], pi + 1
            if use_res:
                res_w, pi = params[pi], pi + 1
            K = cin * ksz
            stats = None
            if use_ln:
                stats = torch.empty((B, 2), device=dev, dtype=torch.float32)
                _hip.call("mx_plane_stats", _hip.ptr(cur), None, B, 1, cin, t_in, float(eps), _hip.ptr(stats), _hip.stream())
            col = torch.empty((K, B * t_out), device=dev, dtype=torch.float32)
            _hip.call("mx_tcn_im2col", _hip.ptr(cur), _hip.ptr(stats), B, cin, t_in, t_out, ksz, dil, stride, _hip.ptr(col),
                      _hip.stream())
            z = torch.empty((B, cout, PITCH), device=dev, dtype=torch.float32)
            wc = w.detach().contiguous()
            _sgemm(_hip.ptr(wc), K, 1, 0, _hip.ptr(col), B * t_out, 1, t_out, _hip.ptr(z), PITCH, 1, cout * PITCH, cout, t_out, K, B)
            r = None
            if use_res:
                r = torch.empty((B, cout, PITCH), device=dev, dtype=torch.float32)
                rw = res_w.detach().contiguous()
                _sgemm(_hip.ptr(rw), cin, 1, 0, _hip.ptr(cur), PITCH, stride, cin * PITCH, _hip.ptr(r), PITCH, 1, cout * PITCH,
                       cout, t_out, cin, B)
            y = torch.empty((B, cout, PITCH), device=dev, dtype=torch.float32)
            _hip.call("mx_tcn_act_fwd", _hip.ptr(z), _hip.ptr(bias.detach().contiguous()),
                      _hip.ptr(slope.detach().contiguous()) if use_act else None, _hip.ptr(r), B, cout, t_out, _hip.ptr(y),
                      _hip.stream())
            saved.append((cur, stats, col, z))
            cur = y
        ctx.cfg, ctx.saved, ctx.need_input_grad = cfg, saved, need_input_grad
        ctx.params = [p.detach() for p in params]
        return cur

    @staticmethod
    def backward(ctx, dy: Tensor):
        cfg, saved, params = ctx.cfg, ctx.saved, ctx.params
        B = dy.size(0)
        dev = dy.device
        grads: List[Optional[Tensor]] = [None] * len(params)
        # parameter offsets per block
        offs, pi = [], 0
        for c in cfg:
            offs.append(pi)
            pi += 2 + int(c[6]) + int(c[7])
        dcur = dy.contiguous()
        per_group = 8
        groups = (B + per_group - 1) // per_group
        for bi in range(len(cfg) - 1, -1, -1):
            cin, cout, ksz, dil, stride, use_ln, use_act, use_res, t_in, t_out, eps = cfg[bi]
            x, stats, col, zb = saved[bi]
            o = offs[bi]
            w = params[o].contiguous()
            slope = params[o + 2].contiguous() if use_act else None
            res_w = params[o + 2 + int(use_act)].contiguous() if use_res else None
            K = cin * ksz
            dz = torch.empty_like(dcur)
            part = torch.empty((B * cout, 2), device=dev, dtype=torch.float32)
            _hip.call("mx_tcn_act_bwd", _hip.ptr(dcur), _hip.ptr(zb), _hip.ptr(slope), B, cout, t_out, _hip.ptr(dz),
                      _hip.ptr(part), _hip.stream())
            sums = _reduce_rows(part, B, 2 * cout).view(cout, 2)
            grads[o + 1] = sums[:, 0].contiguous()
            if use_act:
                grads[o + 2] = sums[:, 1].contiguous()
            # dW[co][K] = sum_{b,t'} dz[b][co][t'] col[K][b, t']
            pw = torch.empty((groups, cout, K), device=dev, dtype=torch.float32)
            _sgemm(_hip.ptr(dz), PITCH, 1, cout * PITCH, _hip.ptr(col), 1, B * t_out, t_out, _hip.ptr(pw), K, 1, cout * K,
                   cout, K, t_out, B, per_group)
            grads[o] = _reduce_rows(pw, groups, cout * K).view(cout, cin, ksz)
            if use_res:
                pr = torch.empty((groups, cout, cin), device=dev, dtype=torch.float32)
                _sgemm(_hip.ptr(dcur), PITCH, 1, cout * PITCH, _hip.ptr(x), stride, PITCH, cin * PITCH, _hip.ptr(pr), cin, 1,
                       cout * cin, cout, cin, t_out, B, per_group)
                grads[o + 2 + int(use_act)] = _reduce_rows(pr, groups, cout * cin).view(cout, cin, 1)
            if bi == 0 and not ctx.need_input_grad:
                dcur = None
                break
            dcol = torch.empty((K, B * t_out), device=dev, dtype=torch.float32)
            _sgemm(_hip.ptr(w), 1, K, 0, _hip.ptr(dz), PITCH, 1, cout * PITCH, _hip.ptr(dcol), B * t_out, 1, t_out, K, t_out,
                   cout, B)
            dxhat = torch.empty((B, cin, PITCH), device=dev, dtype=torch.float32)
            _hip.call("mx_tcn_col2im", _hip.ptr(dcol), B, cin, t_in, t_out, ksz, dil, stride, _hip.ptr(dxhat), _hip.stream())
            dxr = None
            if use_res:
                dxr = torch.zeros((B, cin, PITCH), device=dev, dtype=torch.float32)
                _sgemm(_hip.ptr(res_w), 1, cin, 0, _hip.ptr(dcur), PITCH, 1, cout * PITCH, _hip.ptr(dxr), PITCH, stride,
                       cin * PITCH, cin, t_out, cout, B)
            if use_ln:
                dx = torch.empty((B, cin, PITCH), device=dev, dtype=torch.float32)
                _hip.call("mx_tcn_ln_bwd", _hip.ptr(x), _hip.ptr(dxhat), _hip.ptr(stats), _hip.ptr(dxr), B, cin, t_in,
                          _hip.ptr(dx), _hip.stream())
            else:
                dx = dxhat if dxr is None else dxhat + dxr
            dcur = dx
        return (dcur, None, None, *grads)


def _f32(shape, dev) -> Tensor:
    return torch.empty(shape, device=dev, dtype=torch.float32)


class _GeneralBlockFn(torch.autograd.Function):
    """One TCNBlock (tcn.py:172-192) on dense tensors: x (B, cin, T) -> y (B, cout, To), the new streaming cache and the batch
    statistics of the FiLM BatchNorm.  ``meta``: k, d, s, p, use_ln, eps, causal, cached, film ("none" | "plain" | "bn_train" |
    "bn_eval"), bn_eps, use_act, use_res;  ``params``: conv.weight, conv.bias, [act.weight], [res.weight], [adaptor.weight, .bias]."""

    @staticmethod
    def forward(ctx, x: Tensor, cond: Optional[Tensor], pad_buf: Optional[Tensor], norm_eval: Optional[Tensor], meta: dict, *params: Tensor):
        m = meta
        k, d, s, p = m["k"], m["d"], m["s"], m["p"]
        dev, st = x.device, _hip.stream()
        x = x.contiguous().float()
        B, cin, T = x.shape
        pr = [q.detach().contiguous().float() for q in params]
        w, bias = pr[0], pr[1]
        pi = 2
        slope = res_w = ad_w = ad_b = None
        if m["use_act"]:
            slope, pi = pr[pi], pi + 1
        if m["use_res"]:
            res_w, pi = pr[pi], pi + 1
        if m["film"] != "none":
            ad_w, ad_b = pr[pi], pr[pi + 1]
        cout, K = w.size(0), cin * k
        lnstats = None
        if m["use_ln"]:
            xhat, lnstats = _f32((B, cin, T), dev), _f32((B, 2), dev)
            _hip.call("mx_rowln_fwd", _hip.ptr(x), B, cin * T, float(m["eps"]), _hip.ptr(xhat), _hip.ptr(lnstats), st)
        else:
            xhat = x
        new_pad = _f32((0,), dev)
        pad = 0
        if m["cached"]:                                             # tcn.py:39-46: [cache | input], the tail is the next cache
            pad = (k - 1) * d
            buf = pad_buf if pad_buf.size(0) == B else pad_buf[:1].expand(B, cin, pad)
            xc = torch.cat([buf.to(dev).float(), xhat], dim=-1).contiguous()
            new_pad = xc[..., xc.size(-1) - pad:].clone()
        else:
            xc = xhat
        Tc = xc.size(-1)
        To = (Tc + 2 * p - d * (k - 1) - 1) // s + 1
        assert To >= 1, "the clip is shorter than the receptive field of the block"
        col = _f32((K, B * Tc), dev)
        _hip.call("mx_im2col2d", _hip.ptr(xc), B, cin, 1, Tc, 1, k, 1, d, 0, p, _hip.ptr(col), st)
        z = bias.view(1, cout, 1).expand(B, cout, To).contiguous()                  # the products accumulate onto the bias
        _sgemm(_hip.ptr(w), K, 1, 0, _hip.ptr(col), B * Tc, s, Tc, _hip.ptr(z), To, 1, cout * To, cout, To, K, B, accumulate=1)
        zh = gb = norm = None
        stats = _f32((0,), dev)
        a = z
        if m["film"] != "none":
            cd = ad_w.size(1)
            condc = cond.detach().contiguous().float()
            gb = ad_b.view(1, 2 * cout).expand(B, 2 * cout).contiguous()
            _sgemm(_hip.ptr(condc), cd, 1, 0, _hip.ptr(ad_w), 1, cd, 0, _hip.ptr(gb), 2 * cout, 1, 0, B, 2 * cout, cd, 1, accumulate=1)
            zh = z
            if m["film"] in ("bn_train", "bn_eval"):
                if m["film"] == "bn_train":
                    stats = _f32((cout, 2), dev)
                    _hip.call("mx_chan_stats", _hip.ptr(z), B, cout, To, _hip.ptr(stats), st)
                    norm = torch.stack([stats[:, 0], torch.rsqrt(stats[:, 1] + m["bn_eps"])], dim=1).contiguous()
                else:
                    norm = norm_eval.contiguous().float()
                zh = _f32((B, cout, To), dev)
                _hip.call("mx_chan_norm_fwd", _hip.ptr(z), _hip.ptr(norm), B, cout, To, _hip.ptr(zh), st)
            a = _f32((B, cout, To), dev)
            _hip.call("mx_film_fwd", _hip.ptr(zh), _hip.ptr(gb), B, cout, To, _hip.ptr(a), st)
        r, off = None, 0
        if m["use_res"]:                                            # tcn.py:188-191: 1x1 strided branch of the block INPUT, cropped
            Tr = (T - 1) // s + 1
            assert Tr >= To
            off = 0 if Tr == To else (Tr - 1 - To if m["causal"] else (Tr - To) // 2)
            r = _f32((B, cout, To), dev)
            _sgemm(_hip.ptr(res_w), cin, 1, 0, x.data_ptr() + 4 * off * s, T, s, cin * T, _hip.ptr(r), To, 1, cout * To, cout, To, cin, B)
        y = _f32((B, cout, To), dev)
        _hip.call("mx_prelu_res_fwd", _hip.ptr(a), _hip.ptr(slope), _hip.ptr(r), B, cout, To, _hip.ptr(y), st)
        ctx.meta, ctx.dims = m, (B, cin, T, Tc, To, cout, K, pad, off)
        ctx.tensors = (x, xhat, lnstats, col, a, zh, gb, norm, cond.detach().contiguous().float() if cond is not None else None,
                       w, slope, res_w, ad_w)
        ctx.mark_non_differentiable(new_pad, stats)
        return y, new_pad, stats

    @staticmethod
    def backward(ctx, dy: Tensor, _dpad, _dstats):
        m = ctx.meta
        k, d, s, p = m["k"], m["d"], m["s"], m["p"]
        B, cin, T, Tc, To, cout, K, pad, off = ctx.dims
        x, xhat, lnstats, col, a, zh, gb, norm, cond, w, slope, res_w, ad_w = ctx.tensors
        dev, st = x.device, _hip.stream()
        dy = dy.contiguous().float()
        grads: List[Optional[Tensor]] = []
        d_slope = d_res = d_adw = d_adb = d_cond = None
        if m["use_act"]:
            da, part = _f32((B, cout, To), dev), _f32((B * cout,), dev)
            _hip.call("mx_prelu_res_bwd", _hip.ptr(dy), _hip.ptr(a), _hip.ptr(slope), B, cout, To, _hip.ptr(da), _hip.ptr(part), st)
            d_slope = _reduce_rows(part, B, cout)
        else:
            da = dy
        if m["film"] != "none":
            cd = ad_w.size(1)
            dzh, dgb = _f32((B, cout, To), dev), _f32((B, 2 * cout), dev)
            _hip.call("mx_film_bwd", _hip.ptr(da), _hip.ptr(zh), _hip.ptr(gb), B, cout, To, _hip.ptr(dzh), _hip.ptr(dgb), st)
            d_adw = _f32((2 * cout, cd), dev)
            _sgemm(_hip.ptr(dgb), 1, 2 * cout, 0, _hip.ptr(cond), cd, 1, 0, _hip.ptr(d_adw), cd, 1, 0, 2 * cout, cd, B, 1)
            d_adb = _reduce_rows(dgb, B, 2 * cout)
            if ctx.needs_input_grad[1]:
                d_cond = _f32((B, cd), dev)
                _sgemm(_hip.ptr(dgb), 2 * cout, 1, 0, _hip.ptr(ad_w), cd, 1, 0, _hip.ptr(d_cond), cd, 1, 0, B, cd, 2 * cout, 1)
            if norm is not None:
                dz = _f32((B, cout, To), dev)
                _hip.call("mx_chan_norm_bwd", _hip.ptr(dzh), _hip.ptr(zh), _hip.ptr(norm), B, cout, To, int(m["film"] == "bn_train"),
                          _hip.ptr(dz), st)
            else:
                dz = dzh
        else:
            dz = da
        rs = _f32((B * cout,), dev)
        _hip.call("mx_row_sums", _hip.ptr(dz), B * cout, To, _hip.ptr(rs), st)
        d_bias = _reduce_rows(rs, B, cout)
        pw = _f32((B, cout, K), dev)                                # one partial per clip, reduced in fp64
        _sgemm(_hip.ptr(dz), To, 1, cout * To, _hip.ptr(col), s, B * Tc, Tc, _hip.ptr(pw), K, 1, cout * K, cout, K, To, B)
        d_w = _reduce_rows(pw, B, cout * K).view(cout, cin, k)
        dx = None
        if ctx.needs_input_grad[0]:
            dcol = torch.zeros((K, B * Tc), device=dev, dtype=torch.float32)        # columns no output reads stay zero
            _sgemm(_hip.ptr(w), 1, K, 0, _hip.ptr(dz), To, 1, cout * To, _hip.ptr(dcol), B * Tc, s, Tc, K, To, cout, B)
            dxc = _f32((B, cin, Tc), dev)
            _hip.call("mx_col2im2d", _hip.ptr(dcol), B, cin, 1, Tc, 1, k, 1, d, 0, p, _hip.ptr(dxc), st)
            dxh = dxc[..., pad:].contiguous() if pad else dxc        # the cache is a constant of the call
            if m["use_ln"]:
                dx = _f32((B, cin, T), dev)
                _hip.call("mx_rowln_bwd", _hip.ptr(dxh), _hip.ptr(xhat), _hip.ptr(lnstats), B, cin * T, _hip.ptr(dx), st)
            else:
                dx = dxh
        if m["use_res"]:
            pr = _f32((B, cout, cin), dev)
            _sgemm(_hip.ptr(dy), To, 1, cout * To, x.data_ptr() + 4 * off * s, s, T, cin * T, _hip.ptr(pr), cin, 1, cout * cin, cout, cin,
                   To, B)
            d_res = _reduce_rows(pr, B, cout * cin).view(cout, cin, 1)
            if dx is not None:                                      # d x[b][ci][(off + t') s] += sum over co of res_w[co][ci] dy[b][co][t']
                _sgemm(_hip.ptr(res_w), 1, cin, 0, _hip.ptr(dy), To, 1, cout * To, dx.data_ptr() + 4 * off * s, T, s, cin * T, cin, To,
                       cout, B, accumulate=1)
        grads = [d_w, d_bias]
        if m["use_act"]:
            grads.append(d_slope)
        if m["use_res"]:
            grads.append(d_res)
        if m["film"] != "none":
            grads += [d_adw, d_adb]
        return (dx, d_cond, None, None, None, *grads)


def run_block_general(blk: "TCNBlock", x: Tensor, cond: Optional[Tensor] = None) -> Tensor:
    """tcn.py:172-192 for one block on the general kernels; updates the streaming cache and the BatchNorm running statistics
    (momentum rule of nn.BatchNorm1d: biased variance normalises, the unbiased one is tracked)."""
    assert x.ndim == 3 and x.size(1) == blk.in_ch
    if blk.use_ln:
        assert x.size(2) == blk.temporal_dim                                      # tcn.py:175-177
    conv = blk.conv.conv if blk.is_cached else blk.conv
    film, norm_eval = "none", None
    if blk.film is not None:
        assert cond is not None
        bn = blk.film.bn
        if bn is None:
            film = "plain"
        elif bn.training or not bn.track_running_stats:
            film = "bn_train"
        else:
            film = "bn_eval"
            norm_eval = torch.stack([bn.running_mean, torch.rsqrt(bn.running_var + bn.eps)], dim=1)
    meta = dict(k=blk.kernel_size, d=blk.dilation, s=blk.stride, p=blk.padding, use_ln=blk.use_ln,
                eps=blk.ln.eps if blk.use_ln else 0.0, causal=blk.is_causal, cached=blk.is_cached, film=film,
                bn_eps=blk.film.bn.eps if film.startswith("bn") else 0.0, use_act=blk.use_act, use_res=blk.use_res)
    params = [conv.weight, conv.bias]
    if blk.use_act:
        params.append(blk.act.weight)
    if blk.use_res:
        params.append(blk.res.weight)
    if blk.film is not None:
        params += [blk.film.adaptor.weight, blk.film.adaptor.bias]
    y, new_pad, stats = _GeneralBlockFn.apply(x, cond, blk.conv.pad.pad_buf if blk.is_cached else None, norm_eval, meta, *params)
    if blk.is_cached:
        blk.conv.pad.pad_buf = new_pad
    if film == "bn_train" and blk.film.bn.track_running_stats:
        bn, n = blk.film.bn, y.size(0) * y.size(2)
        with torch.no_grad():
            bn.num_batches_tracked += 1
            mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
            bn.running_mean.mul_(1.0 - mom).add_(stats[:, 0], alpha=mom)
            bn.running_var.mul_(1.0 - mom).add_(stats[:, 1] * (n / max(n - 1, 1)), alpha=mom)
    return y


class FiLM(nn.Module):
    """tcn.py:82-103: parameter / buffer holder (``bn.running_mean`` / ``running_var`` / ``num_batches_tracked``,
    ``adaptor.weight`` / ``.bias``); the arithmetic runs inside ``_GeneralBlockFn``.  The BatchNorm1d has batch statistics:
    like the reference's (plain ``nn.BatchNorm1d``, no SyncBatchNorm) they are per process under DDP."""

    def __init__(self, cond_dim: int, num_features: int, use_bn: bool = True) -> None:
        super().__init__()
        self.num_features, self.use_bn = num_features, use_bn
        self.bn = nn.BatchNorm1d(num_features, affine=False) if use_bn else None
        self.adaptor = nn.Linear(cond_dim, 2 * num_features)


class PaddingCached(nn.Module):
    """tcn.py:31-47: the streaming cache (buffer ``pad_buf``: the last ``padding`` input frames of the previous call)."""

    def __init__(self, n_ch: int, padding: int) -> None:
        super().__init__()
        self.n_ch, self.padding = n_ch, padding
        self.register_buffer("pad_buf", torch.zeros((1, n_ch, padding)))


class Conv1dCached(nn.Module):
    """tcn.py:50-79: parameter holder with the reference's keys (``pad.pad_buf``, ``conv.weight``, ``conv.bias``)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, stride: int, padding: int = 0, dilation: int = 1,
                 bias: bool = True) -> None:
        super().__init__()
        assert padding == 0 and bias
        self.pad = PaddingCached(in_channels, (kernel_size - 1) * dilation)
        self.conv = nn.Conv1d(in_channels, out_channels, (kernel_size,), (stride,), padding=0, dilation=(dilation,), bias=True)


class TCNBlock(nn.Module):
    """tcn.py:106-195 (parameter holder; ``TCN.forward`` / ``run_blocks`` run the kernels)."""

    def __init__(self, in_ch: int, out_ch: int, kernel_size: int = 3, dilation: int = 1, stride: int = 1,
                 padding: Optional[int] = 0, use_ln: bool = False, temporal_dim: Optional[int] = None, use_act: bool = True,
                 use_res: bool = True, cond_dim: int = 0, use_film_bn: bool = True, is_causal: bool = True,
                 is_cached: bool = False) -> None:
        super().__init__()
        if is_causal:
            assert padding == 0, "If the TCN is causal, padding must be 0"
        if is_cached:
            assert is_causal, "If the TCN is streaming, it must be causal"
        if padding is None:
            padding = kernel_size // 2 * dilation
        if 2 * padding > dilation * (kernel_size - 1):
            raise NotImplementedError("padding beyond 'same' (an output longer than its input) is not covered by the HIP path")
        self.in_ch, self.out_ch, self.kernel_size, self.dilation, self.stride = in_ch, out_ch, kernel_size, dilation, stride
        self.use_ln, self.temporal_dim, self.use_act, self.use_res = use_ln, temporal_dim, use_act, use_res
        self.cond_dim, self.use_film_bn, self.is_causal, self.is_cached, self.padding = cond_dim, use_film_bn, is_causal, is_cached, padding
        self.crop_fn = causal_crop if is_causal else center_crop
        # the 352-column plane kernels (_TCNStack) cover the non-causal, automatically padded, unconditioned block
        self.fast = (not is_causal and not is_cached and cond_dim == 0 and kernel_size % 2 == 1
                     and padding == kernel_size // 2 * dilation)
        self.ln = None
        if use_ln:
            assert temporal_dim is not None and temporal_dim > 0
            self.ln = nn.LayerNorm([in_ch, temporal_dim], elementwise_affine=False)
        self.act = nn.PReLU(out_ch) if use_act else None
        if is_cached:
            self.conv = Conv1dCached(in_ch, out_ch, kernel_size, stride=stride, padding=padding, dilation=dilation, bias=True)
        else:
            self.conv = nn.Conv1d(in_ch, out_ch, kernel_size, stride=stride, padding=padding, dilation=dilation, bias=True)
        self.res = nn.Conv1d(in_ch, out_ch, kernel_size=(1,), stride=(stride,), bias=False) if use_res else None
        self.film = FiLM(cond_dim, out_ch, use_bn=use_film_bn) if cond_dim > 0 else None

    def is_conditional(self) -> bool:
        return self.cond_dim > 0

    def out_len(self, t_in: int) -> int:
        return (t_in + 2 * self.padding - self.dilation * (self.kernel_size - 1) - 1) // self.stride + 1

    def forward(self, x: Tensor, cond: Optional[Tensor] = None) -> Tensor:
        return run_blocks([self], x, cond)


def run_blocks(blocks: List[TCNBlock], x: Tensor, cond: Optional[Tensor] = None) -> Tensor:
    assert x.ndim == 3
    B, C, T = x.shape
    if T > PITCH or not all(b.fast for b in blocks):
        for blk in blocks:
            x = run_block_general(blk, x, cond)
        return x
    planes = torch.zeros((B, C, PITCH), device=x.device, dtype=torch.float32)
    planes[:, :, :T] = x
    t_out = run_blocks_planes(blocks, planes, T, x.requires_grad)
    return t_out[0][:, :, :t_out[1]]


def run_blocks_planes(blocks: List[TCNBlock], planes: Tensor, T: int, need_input_grad: bool = False) -> Tuple[Tensor, int]:
    """planes (B, C, 352) with T valid frames -> ((B, C_out, 352), T_out)"""
    cfg, params, t = [], [], T
    for blk in blocks:
        assert planes.size(1) == blk.in_ch or cfg, "input channels do not match the first block"
        if blk.use_ln:
            assert t == blk.temporal_dim, (t, blk.temporal_dim)          # tcn.py:175-177
        to = blk.out_len(t)
        if blk.use_res:
            assert (t - 1) // blk.stride + 1 == to                        # the 1x1 branch needs no crop (tcn.py:188-190)
        cfg.append((blk.in_ch, blk.out_ch, blk.kernel_size, blk.dilation, blk.stride, blk.use_ln, blk.use_act, blk.use_res,
                    t, to, blk.ln.eps if blk.use_ln else 0.0))
        params += [blk.conv.weight, blk.conv.bias]
        if blk.use_act:
            params.append(blk.act.weight)
        if blk.use_res:
            params.append(blk.res.weight)
        t = to
    return _TCNStack.apply(planes, tuple(cfg), need_input_grad, *params), t


class TCN(nn.Module):
    """tcn.py:198-302."""

    def __init__(self, out_channels: List[int], dilations: Optional[List[int]] = None, in_ch: int = 1, kernel_size: int = 13,
                 strides: Optional[List[int]] = None, padding: Optional[int] = 0, use_ln: bool = False,
                 temporal_dims: Optional[List[int]] = None, use_act: bool = True, use_res: bool = True, cond_dim: int = 0,
                 use_film_bn: bool = False, is_causal: bool = True, is_cached: bool = False) -> None:
        super().__init__()
        self.out_channels, self.in_ch, self.out_ch = out_channels, in_ch, out_channels[-1]
        self.kernel_size, self.padding, self.use_ln, self.temporal_dims = kernel_size, padding, use_ln, temporal_dims
        self.use_act, self.use_res, self.cond_dim, self.use_film_bn = use_act, use_res, cond_dim, use_film_bn
        self.is_causal, self.is_cached = is_causal, is_cached
        self.crop_fn = causal_crop if is_causal else center_crop
        self.n_blocks = len(out_channels)
        if dilations is None:
            dilations = [4 ** idx for idx in range(self.n_blocks)]
            log.info(f"Setting dilations automatically to: {dilations}")
        assert len(dilations) == self.n_blocks
        self.dilations = dilations
        if strides is None:
            strides = [1] * self.n_blocks
            log.info(f"Setting strides automatically to: {strides}")
        assert len(strides) == self.n_blocks
        self.strides = strides
        if use_ln:
            assert temporal_dims is not None and len(temporal_dims) == self.n_blocks
        self.blocks = nn.ModuleList()
        block_out_ch = None
        for idx, (curr_out_ch, dil, stride) in enumerate(zip(out_channels, dilations, strides)):
            block_in_ch = in_ch if idx == 0 else block_out_ch
            block_out_ch = curr_out_ch
            temp_dim = temporal_dims[idx] if temporal_dims is not None else None
            self.blocks.append(TCNBlock(block_in_ch, block_out_ch, kernel_size, dil, stride, padding, use_ln, temp_dim,
                                        use_act, use_res, cond_dim, use_film_bn, is_causal, is_cached))

    def is_conditional(self) -> bool:
        return self.cond_dim > 0

    def forward(self, x: Tensor, cond: Optional[Tensor] = None) -> Tensor:
        assert x.ndim == 3                      # (batch_size, in_ch, samples)
        if self.is_conditional():
            assert cond is not None
            assert cond.shape == (x.size(0), self.cond_dim)  # (batch_size, cond_dim)
        return run_blocks(list(self.blocks), x, cond)

    def forward_planes(self, planes: Tensor, T: int) -> Tuple[Tensor, int]:
        return run_blocks_planes(list(self.blocks), planes, T)

    def calc_receptive_field(self) -> int:
        """tcn.py:295-302."""
        assert all(_ == 1 for _ in self.strides)
        assert self.dilations[0] == 1
        rf = self.kernel_size
        for dil in self.dilations[1:]:
            rf = rf + ((self.kernel_size - 1) * dil)
        return rf
