"""Golden vectors for the TCN stack from the REAL reference module (mod_extraction/tcn.py is importable as it is):
forward outputs and every gradient of three small configurations (plain, dilated with LayerNorm, strided as in
SpectralDSTCN).  Only the vectors are committed.

    cd tests/golden && PYTHONDONTWRITEBYTECODE=1 python make_golden_tcn.py
"""
import os
import sys

import numpy as np
import torch as tr

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

CASES = [
    dict(name="a", in_ch=20, out_channels=[8, 8, 8], dilations=[1, 2, 4], kernel_size=5, strides=None, use_ln=True, T=50),
    dict(name="b", in_ch=17, out_channels=[32, 32], dilations=[1, 16], kernel_size=13, strides=None, use_ln=True, T=120),
    dict(name="c", in_ch=12, out_channels=[16, 16, 16], dilations=[1, 2, 4], kernel_size=13, strides=[2, 2, 2], use_ln=True, T=45),
    dict(name="d", in_ch=7, out_channels=[8, 8], dilations=[1, 3], kernel_size=3, strides=None, use_ln=False, T=40),
]


def temporal_dims(T, strides, n):
    out, cur = [T], T
    for s in (strides or [1] * n)[:-1]:
        cur = -(-cur // s)
        out.append(cur)
    return out


def main():
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    from mod_extraction import tcn as rtcn
    out = {}
    for ci, c in enumerate(CASES):
        tr.manual_seed(100 + ci)
        n = len(c["out_channels"])
        net = rtcn.TCN(c["out_channels"], c["dilations"], c["in_ch"], c["kernel_size"], c["strides"], padding=None,
                       use_ln=c["use_ln"], temporal_dims=temporal_dims(c["T"], c["strides"], n), use_res=True, is_causal=False)
        with tr.no_grad():
            for b in net.blocks:
                b.act.weight.uniform_(0.05, 0.45)
        x = tr.randn(3, c["in_ch"], c["T"], requires_grad=True)
        y = net(x)
        w = tr.linspace(0.5, 1.5, y.numel()).view_as(y)
        (y * w).sum().backward()
        k = c["name"]
        out[f"{k}_x"], out[f"{k}_y"], out[f"{k}_dx"] = x.detach().numpy(), y.detach().numpy(), x.grad.numpy()
        for name, p in net.named_parameters():
            out[f"{k}_p_{name}"] = p.detach().numpy()
            out[f"{k}_g_{name}"] = p.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "tcn.npz"), **out)
    print("wrote tcn.npz", len(out), "arrays", sum(v.nbytes for v in out.values()) // 1024, "KiB")


if __name__ == "__main__":
    main()
