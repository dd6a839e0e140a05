"""Host mirror of mod_extraction/tcn.py: ``TCN`` / ``TCNBlock`` with the reference's constructor arguments, module tree
and state-dict keys (``blocks.<i>.conv.weight``, ``.conv.bias``, ``.act.weight``, ``.res.weight``); the arithmetic runs in
``csrc/tcn.hip`` (LayerNorm statistics -> im2col gather -> fp32 GEMM on the matrix cores -> bias / PReLU / residual, and
the matching backward) as ONE autograd node over the whole stack.

Built for what ``SpectralTCN`` / ``SpectralDSTCN`` (models.py:72-125,218-289) use: non-causal blocks with automatic
padding (``padding=None``), optional LayerNorm / PReLU / residual, any dilation and stride, at most 352 frames.  The
streaming variants (``is_causal`` / ``is_cached``: ``Conv1dCached`` / ``PaddingCached``, tcn.py:34-79) and FiLM
conditioning (tcn.py:82-103) are used by no model of the reference and raise ``NotImplementedError``.
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


class FiLM(nn.Module):
    """tcn.py:82-103 -- not used by any model of the reference; its BatchNorm1d carries batch statistics (the one op of
    the tree that would need an all-reduce under DDP)."""

    def __init__(self, cond_dim: int, num_features: int, use_bn: bool = True) -> None:
        super().__init__()
        raise NotImplementedError("FiLM conditioning (tcn.py:82-103) is not used by SpectralTCN / SpectralDSTCN")


class TCNBlock(nn.Module):
    """tcn.py:106-195 (parameter holder; ``TCN.forward`` runs the whole stack in the HIP kernels)."""

    def __init__(self, in_ch: int, out_ch: int, kernel_size: int = 3, dilation: int = 1, stride: int = 1,
                 padding: Optional[int] = 0, use_ln: bool = False, temporal_dim: Optional[int] = None, use_act: bool = True,
                 use_res: bool = True, cond_dim: int = 0, use_film_bn: bool = True, is_causal: bool = True,
                 is_cached: bool = False) -> None:
        super().__init__()
        if is_causal or is_cached:
            raise NotImplementedError("causal / cached (streaming) TCN blocks are not used by SpectralTCN / SpectralDSTCN")
        if cond_dim > 0:
            raise NotImplementedError("FiLM conditioning is not used by SpectralTCN / SpectralDSTCN")
        if padding is None:
            padding = kernel_size // 2 * dilation
        if padding != kernel_size // 2 * dilation or kernel_size % 2 != 1:
            raise NotImplementedError("the HIP path covers odd kernels with padding = (kernel_size // 2) * dilation")
        self.in_ch, self.out_ch, self.kernel_size, self.dilation, self.stride = in_ch, out_ch, kernel_size, dilation, stride
        self.use_ln, self.temporal_dim, self.use_act, self.use_res = use_ln, temporal_dim, use_act, use_res
        self.cond_dim, self.use_film_bn, self.is_causal, self.is_cached, self.padding = cond_dim, use_film_bn, is_causal, is_cached, padding
        self.crop_fn = center_crop
        self.ln = None
        if use_ln:
            assert temporal_dim is not None and temporal_dim > 0
            self.ln = nn.LayerNorm([in_ch, temporal_dim], elementwise_affine=False)
        self.act = nn.PReLU(out_ch) if use_act else None
        self.conv = nn.Conv1d(in_ch, out_ch, kernel_size, stride=stride, padding=padding, dilation=dilation, bias=True)
        self.res = nn.Conv1d(in_ch, out_ch, kernel_size=(1,), stride=(stride,), bias=False) if use_res else None
        self.film = None

    def is_conditional(self) -> bool:
        return self.cond_dim > 0

    def out_len(self, t_in: int) -> int:
        return (t_in + 2 * self.padding - self.dilation * (self.kernel_size - 1) - 1) // self.stride + 1

    def forward(self, x: Tensor, cond: Optional[Tensor] = None) -> Tensor:
        return run_blocks([self], x)


def run_blocks(blocks: List[TCNBlock], x: Tensor) -> Tensor:
    assert x.ndim == 3
    B, C, T = x.shape
    if T > PITCH:
        raise NotImplementedError(f"the TCN kernels hold at most {PITCH} frames per clip")
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
        return run_blocks(list(self.blocks), x)

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
