"""Oracle restatement of the REST of mod_extraction/tcn.py -- what oracle/tcn.py leaves out: explicit padding with the
causal / centre crop of the residual branch (tcn.py:14-29,130-134,188-191), the cached (streaming) convolution
(tcn.py:31-79) and FiLM conditioning with or without its affine-free BatchNorm1d (tcn.py:82-103,172-184) -- TEST
INFRASTRUCTURE ONLY, torch fp32 on the CPU.  Module names follow the reference (``blocks.<i>.conv`` -- ``.conv.conv`` and
``.conv.pad.pad_buf`` when cached --, ``.act``, ``.res``, ``.film.bn``, ``.film.adaptor``) so state dicts interchange.
Pinned by tests/golden/make_golden_tcn_general.py -> tcn_general.npz (outputs, gradients, streaming state and BatchNorm
running statistics of the REAL ``tcn.TCN``)."""
from typing import List, Optional

import torch
import torch.nn.functional as F
from torch import Tensor, nn


def crop_centre(x: Tensor, length: int) -> Tensor:                      # tcn.py:14-20
    extra = x.size(-1) - length
    assert extra >= 0
    return x if extra == 0 else x[..., extra // 2:extra // 2 + length]


def crop_causal(x: Tensor, length: int) -> Tensor:                      # tcn.py:23-29: ends ONE frame before the last
    if x.size(-1) == length:
        return x
    assert x.size(-1) > length
    stop = x.size(-1) - 1
    return x[..., stop - length:stop]


class _Pad(nn.Module):
    def __init__(self, n_ch: int, padding: int) -> None:
        super().__init__()
        self.padding = padding
        self.register_buffer("pad_buf", torch.zeros((1, n_ch, padding)))


class CachedConv(nn.Module):
    """tcn.py:50-79: the last (k - 1) d input frames of the previous call stand in for the left padding."""

    def __init__(self, cin: int, cout: int, k: int, stride: int, dilation: int) -> None:
        super().__init__()
        self.pad = _Pad(cin, (k - 1) * dilation)
        self.conv = nn.Conv1d(cin, cout, (k,), (stride,), padding=0, dilation=(dilation,), bias=True)

    def forward(self, x: Tensor) -> Tensor:
        buf = self.pad.pad_buf
        if x.size(0) > buf.size(0):
            buf = buf.repeat(x.size(0), 1, 1)
        x = torch.cat([buf, x], dim=-1)
        self.pad.pad_buf = x[..., x.size(-1) - self.pad.padding:]
        return self.conv(x)


class FiLM(nn.Module):
    def __init__(self, cond_dim: int, n_feat: int, use_bn: bool) -> None:
        super().__init__()
        self.bn = nn.BatchNorm1d(n_feat, affine=False) if use_bn else None
        self.adaptor = nn.Linear(cond_dim, 2 * n_feat)

    def forward(self, x: Tensor, cond: Tensor) -> Tensor:              # tcn.py:95-103
        gain, shift = self.adaptor(cond).chunk(2, dim=-1)
        if self.bn is not None:
            x = self.bn(x)
        return x * gain.unsqueeze(-1) + shift.unsqueeze(-1)


class TCNBlock(nn.Module):
    def __init__(self, in_ch: int, out_ch: int, kernel_size: int, dilation: int, stride: int, padding: Optional[int], use_ln: bool,
                 temporal_dim: Optional[int], use_act: bool, use_res: bool, cond_dim: int, use_film_bn: bool, is_causal: bool,
                 is_cached: bool) -> None:
        super().__init__()
        assert not is_causal or padding == 0
        assert not is_cached or is_causal
        self.in_ch, self.temporal_dim, self.use_ln, self.is_causal = in_ch, temporal_dim, use_ln, is_causal
        if padding is None:
            padding = kernel_size // 2 * dilation
        self.act = nn.PReLU(out_ch) if use_act else None               # registration order of tcn.py:163-184
        if is_cached:
            self.conv = CachedConv(in_ch, out_ch, kernel_size, stride, dilation)
        else:
            self.conv = nn.Conv1d(in_ch, out_ch, kernel_size, stride=stride, padding=padding, dilation=dilation, bias=True)
        self.res = nn.Conv1d(in_ch, out_ch, kernel_size=(1,), stride=(stride,), bias=False) if use_res else None
        self.film = FiLM(cond_dim, out_ch, use_film_bn) if cond_dim > 0 else None

    def forward(self, x: Tensor, cond: Optional[Tensor] = None) -> Tensor:
        x_in = x
        if self.use_ln:
            assert x.shape[1:] == (self.in_ch, self.temporal_dim)
            x = F.layer_norm(x, [self.in_ch, self.temporal_dim], eps=1e-5)
        x = self.conv(x)
        if self.film is not None:
            x = self.film(x, cond)
        if self.act is not None:
            x = self.act(x)
        if self.res is not None:
            r = self.res(x_in)
            x = x + (crop_causal if self.is_causal else crop_centre)(r, x.size(-1))
        return x


class TCN(nn.Module):
    def __init__(self, out_channels: List[int], dilations: Optional[List[int]] = None, in_ch: int = 1, kernel_size: int = 13,
                 strides: Optional[List[int]] = None, padding: Optional[int] = 0, use_ln: bool = False,
                 temporal_dims: Optional[List[int]] = None, use_act: bool = True, use_res: bool = True, cond_dim: int = 0,
                 use_film_bn: bool = False, is_causal: bool = True, is_cached: bool = False) -> None:
        super().__init__()
        n = len(out_channels)
        dilations = [4 ** i for i in range(n)] if dilations is None else dilations
        strides = [1] * n if strides is None else strides
        self.cond_dim = cond_dim
        self.blocks = nn.ModuleList()
        c = in_ch
        for i, (oc, d, s) in enumerate(zip(out_channels, dilations, strides)):
            self.blocks.append(TCNBlock(c, oc, kernel_size, d, s, padding, use_ln, temporal_dims[i] if temporal_dims else None,
                                        use_act, use_res, cond_dim, use_film_bn, is_causal, is_cached))
            c = oc

    def forward(self, x: Tensor, cond: Optional[Tensor] = None) -> Tensor:
        assert (cond is not None and cond.shape == (x.size(0), self.cond_dim)) or self.cond_dim == 0
        for b in self.blocks:
            x = b(x, cond)
        return x
