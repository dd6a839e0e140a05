"""Golden vectors of the REAL reference ``mod_extraction.tcn.TCN`` (importable as it is) for the variants outside
SpectralTCN / SpectralDSTCN: explicit padding with the causal / centre crop of the residual branch, the cached (streaming)
convolution over two consecutive calls, FiLM conditioning with and without its BatchNorm1d (training mode: batch statistics
and the updated running statistics; evaluation mode: the running statistics).  Only the vectors are committed.

    cd tests/golden && PYTHONDONTWRITEBYTECODE=1 python make_golden_tcn_general.py
"""
import os
import sys

import numpy as np
import torch as tr

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

# kw = constructor keywords of tcn.TCN; T = frames per call; calls = consecutive forward calls (streaming); mode = train / eval
CASES = [
    dict(name="causal", T=70, calls=1, mode="train",
         kw=dict(out_channels=[6, 9, 9], dilations=[1, 2, 4], in_ch=5, kernel_size=5, padding=0, is_causal=True)),
    dict(name="causal_defaults", T=400, calls=1, mode="train",                  # the class defaults: dilations 4^i, 13 taps, causal
         kw=dict(out_channels=[4, 4, 4], in_ch=3)),
    dict(name="padded", T=64, calls=1, mode="train",                            # padding 1 < "same": centre crop of the residual
         kw=dict(out_channels=[8, 8], dilations=[1, 3], in_ch=4, kernel_size=5, padding=1, is_causal=False)),
    dict(name="even_kernel", T=50, calls=1, mode="train",
         kw=dict(out_channels=[7], dilations=[2], in_ch=3, kernel_size=4, padding=3, is_causal=False, use_res=False)),
    dict(name="strided_causal", T=90, calls=1, mode="train",
         kw=dict(out_channels=[6, 6], dilations=[1, 2], in_ch=4, kernel_size=3, strides=[2, 3], padding=0, is_causal=True,
                 use_res=False)),
    # (forward only: the reference keeps the autograd graph of the previous call in its cache, a second backward raises)
    dict(name="cached", T=37, calls=3, mode="eval", grad=False,
         kw=dict(out_channels=[6, 6], dilations=[1, 3], in_ch=2, kernel_size=5, padding=0, is_causal=True, is_cached=True)),
    dict(name="film", T=48, calls=1, mode="train",
         kw=dict(out_channels=[8, 8], dilations=[1, 2], in_ch=3, kernel_size=5, padding=None, is_causal=False, cond_dim=4,
                 use_film_bn=False)),
    dict(name="film_bn_train", T=48, calls=2, mode="train",
         kw=dict(out_channels=[8, 5], dilations=[1, 2], in_ch=3, kernel_size=5, padding=None, is_causal=False, cond_dim=3,
                 use_film_bn=True)),
    dict(name="film_bn_eval", T=48, calls=1, mode="eval",
         kw=dict(out_channels=[8, 5], dilations=[1, 2], in_ch=3, kernel_size=5, padding=None, is_causal=False, cond_dim=3,
                 use_film_bn=True)),
]


def randomise(net):
    """away from the trivial initial values (PReLU 0.25, running mean 0 / var 1)"""
    with tr.no_grad():
        for b in net.blocks:
            if b.act is not None:
                b.act.weight.uniform_(0.05, 0.45)
            if b.film is not None and b.film.bn is not None:
                b.film.bn.running_mean.uniform_(-0.3, 0.3)
                b.film.bn.running_var.uniform_(0.5, 1.5)


def main():
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    from mod_extraction import tcn as rtcn
    out = {}
    for ci, c in enumerate(CASES):
        tr.manual_seed(500 + ci)
        net = rtcn.TCN(**c["kw"])
        randomise(net)
        net.train(c["mode"] == "train")
        k = c["name"]
        for name, t in net.state_dict().items():
            out[f"{k}_p_{name}"] = t.detach().numpy().copy()
        cond_dim = c["kw"].get("cond_dim", 0)
        for call in range(c["calls"]):
            want_grad = c.get("grad", True)
            x = tr.randn(3, c["kw"]["in_ch"], c["T"], requires_grad=want_grad)
            cond = tr.randn(3, cond_dim, requires_grad=True) if cond_dim else None
            net.zero_grad()
            with tr.set_grad_enabled(want_grad):
                y = net(x, cond)
            out[f"{k}_x{call}"], out[f"{k}_y{call}"] = x.detach().numpy(), y.detach().numpy()
            if cond is not None:
                out[f"{k}_c{call}"] = cond.detach().numpy()
            if want_grad:
                w = tr.linspace(0.5, 1.5, y.numel()).view_as(y)
                (y * w).sum().backward()
                out[f"{k}_dx{call}"] = x.grad.numpy()
                if cond is not None:
                    out[f"{k}_dc{call}"] = cond.grad.numpy()
                for name, p in net.named_parameters():
                    out[f"{k}_g{call}_{name}"] = p.grad.numpy().copy()
            for name, t in net.named_buffers():                         # streaming cache / running statistics AFTER the call
                out[f"{k}_b{call}_{name}"] = t.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, "tcn_general.npz"), **out)
    print("wrote tcn_general.npz", len(out), "arrays", sum(v.nbytes for v in out.values()) // 1024, "KiB")


if __name__ == "__main__":
    main()
