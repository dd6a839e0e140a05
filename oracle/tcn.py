"""Oracle restatement of mod_extraction/tcn.py:106-302 for the configurations the product supports (non-causal blocks,
automatic padding, optional LayerNorm / PReLU / 1x1 residual, dilation, stride) -- TEST INFRASTRUCTURE ONLY, torch fp32
on the CPU.  Module names follow the reference (``blocks.<i>.conv`` / ``.act`` / ``.res``) so state dicts interchange.
Pinned (2e-6; bit-identical at equal thread count) by tests/golden/make_golden_tcn.py -> tcn.npz (outputs and gradients of the REAL ``tcn.TCN``)."""
from typing import List, Optional

import torch
import torch.nn.functional as F
from torch import Tensor, nn


class TCNBlock(nn.Module):
    def __init__(self, in_ch: int, out_ch: int, kernel_size: int, dilation: int, stride: int, use_ln: bool,
                 temporal_dim: Optional[int], use_act: bool, use_res: bool) -> None:
        super().__init__()
        self.in_ch, self.temporal_dim, self.use_ln = in_ch, temporal_dim, use_ln
        pad = kernel_size // 2 * dilation                                   # tcn.py:153-155 (padding=None)
        self.act = nn.PReLU(out_ch) if use_act else None                    # registration order of tcn.py:163-183
        self.conv = nn.Conv1d(in_ch, out_ch, kernel_size, stride=stride, padding=pad, dilation=dilation, bias=True)
        self.res = nn.Conv1d(in_ch, out_ch, kernel_size=(1,), stride=(stride,), bias=False) if use_res else None

    def forward(self, x: Tensor) -> Tensor:
        x_in = x
        if self.use_ln:                                                     # tcn.py:174-178
            assert x.shape[1:] == (self.in_ch, self.temporal_dim)
            x = F.layer_norm(x, [self.in_ch, self.temporal_dim], eps=1e-5)
        x = self.conv(x)
        if self.act is not None:
            x = self.act(x)
        if self.res is not None:                                            # tcn.py:188-191 (centre crop: a no-op here)
            r = self.res(x_in)
            assert r.size(-1) == x.size(-1)
            x = x + r
        return x


class TCN(nn.Module):
    def __init__(self, out_channels: List[int], dilations: List[int], in_ch: int, kernel_size: int = 13,
                 strides: Optional[List[int]] = None, use_ln: bool = False, temporal_dims: Optional[List[int]] = None,
                 use_act: bool = True, use_res: bool = True) -> None:
        super().__init__()
        strides = strides or [1] * len(out_channels)
        self.blocks = nn.ModuleList()
        c = in_ch
        for i, (oc, d, s) in enumerate(zip(out_channels, dilations, strides)):
            self.blocks.append(TCNBlock(c, oc, kernel_size, d, s, use_ln, temporal_dims[i] if temporal_dims else None,
                                        use_act, use_res))
            c = oc

    def forward(self, x: Tensor) -> Tensor:
        for b in self.blocks:
            x = b(x)
        return x
