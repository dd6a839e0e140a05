"""The TCN extractors (tcn.py:106-302, models.py:72-125,218-289).
CPU: the oracle restatement against outputs and gradients of the REAL ``tcn.TCN`` (tests/golden/make_golden_tcn.py),
to 2e-6 (same torch operators; thread-count dependent summation order); state-dict keys of the product mirrors.  GPU (-m gpu): the HIP stack against the same vectors (1e-5 of each
tensor's max, gradients 2e-5) and the full SpectralTCN / SpectralDSTCN models against the oracle models."""
import os

import numpy as np
import pytest
import torch

from tests.golden.make_golden_tcn import CASES, temporal_dims


def _load(golden_dir):
    return np.load(os.path.join(golden_dir, "tcn.npz"))


def _build(mod, c):
    n = len(c["out_channels"])
    return dict(out_channels=c["out_channels"], dilations=c["dilations"], in_ch=c["in_ch"], kernel_size=c["kernel_size"],
                strides=c["strides"], use_ln=c["use_ln"], temporal_dims=temporal_dims(c["T"], c["strides"], n), use_res=True)


def _rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def test_oracle_tcn_matches_the_reference(golden_dir):
    from oracle import tcn as otcn
    g = _load(golden_dir)
    for c in CASES:
        k = c["name"]
        net = otcn.TCN(**_build(otcn, c))
        sd = {n[len(f"{k}_p_"):]: torch.from_numpy(g[n]) for n in g.files if n.startswith(f"{k}_p_")}
        net.load_state_dict(sd, strict=True)
        x = torch.from_numpy(g[f"{k}_x"]).requires_grad_(True)
        y = net(x)
        # same torch CPU operators on both sides: identical up to the summation order of the host's thread count
        # (bit-identical on the 8-core container that generated the vectors; <= 2e-6 of the tensor's max elsewhere)
        assert _rel(y.detach().numpy(), g[f"{k}_y"]) < 2e-6, k
        w = torch.linspace(0.5, 1.5, y.numel()).view_as(y)
        (y * w).sum().backward()
        assert _rel(x.grad.numpy(), g[f"{k}_dx"]) < 2e-6, k
        for n, p in net.named_parameters():
            assert _rel(p.grad.numpy(), g[f"{k}_g_{n}"]) < 2e-6, (k, n)


def test_product_state_dict_keys_and_unsupported_variants():
    from mod_extraction_amd import models, tcn
    net = tcn.TCN([8, 8], [1, 2], 5, 3, None, None, True, [30, 30], is_causal=False)
    assert list(net.state_dict().keys()) == ["blocks.0.act.weight", "blocks.0.conv.weight", "blocks.0.conv.bias",
                                             "blocks.0.res.weight", "blocks.1.act.weight", "blocks.1.conv.weight",
                                             "blocks.1.conv.bias", "blocks.1.res.weight"]
    assert net.calc_receptive_field() == 3 + 2 * 2
    assert all(b.fast for b in net.blocks)                          # the SpectralTCN family keeps the plane kernels
    causal = tcn.TCN([8], [1], 5, 3)                               # the reference's default is causal: the general kernels
    assert not causal.blocks[0].fast and causal.blocks[0].crop_fn is tcn.causal_crop
    film = tcn.TCN([8], [1], 5, 3, None, None, cond_dim=3, use_film_bn=True, is_causal=False)
    assert [k for k in film.state_dict() if "film" in k] == [
        "blocks.0.film.bn.running_mean", "blocks.0.film.bn.running_var", "blocks.0.film.bn.num_batches_tracked",
        "blocks.0.film.adaptor.weight", "blocks.0.film.adaptor.bias"]
    cached = tcn.TCN([8], [2], 5, 3, is_cached=True)
    assert "blocks.0.conv.pad.pad_buf" in cached.state_dict() and cached.state_dict()["blocks.0.conv.pad.pad_buf"].shape == (1, 5, 4)
    with pytest.raises(NotImplementedError):                       # an output longer than its input
        tcn.TCN([8], [1], 5, 3, None, 5, is_causal=False)
    m = models.SpectralTCN(n_samples=22272, out_channels=[16, 16], dilations=[1, 2])
    keys = list(m.state_dict().keys())
    assert keys[0] == "spectrogram.window" and "tcn.blocks.1.res.weight" in keys and keys[-2:] == ["output.weight", "output.bias"]
    assert m.receptive_field == 13 + 12 * 2
    d = models.SpectralDSTCN(n_samples=22272, out_channels=[16, 16, 16], dilations=[1, 2, 4])
    assert [b.temporal_dim for b in d.tcn.blocks] == [88, 44, 22] and d.output.out_features == 2
    # the oracle models use the same keys: weights interchange
    from oracle import models as om
    assert list(om.SpectralTCN(n_samples=22272, out_channels=[16, 16], dilations=[1, 2]).state_dict().keys()) == keys
    assert list(om.SpectralDSTCN(n_samples=22272, out_channels=[16, 16, 16], dilations=[1, 2, 4]).state_dict().keys()) == \
        list(d.state_dict().keys())
    assert center_crop_ok()


def center_crop_ok():
    from mod_extraction_amd import tcn
    x = torch.arange(10.0).view(1, 1, 10)
    return tcn.center_crop(x, 4).tolist() == [[[3.0, 4.0, 5.0, 6.0]]] and tcn.causal_crop(x, 4).tolist() == [[[5.0, 6.0, 7.0, 8.0]]]


@pytest.mark.gpu
def test_tcn_stack_vs_reference_golden(golden_dir, dev):
    from mod_extraction_amd import tcn
    g = _load(golden_dir)
    for c in CASES:
        k = c["name"]
        n = len(c["out_channels"])
        net = tcn.TCN(c["out_channels"], c["dilations"], c["in_ch"], c["kernel_size"], c["strides"], padding=None,
                      use_ln=c["use_ln"], temporal_dims=temporal_dims(c["T"], c["strides"], n), use_res=True, is_causal=False)
        sd = {nm[len(f"{k}_p_"):]: torch.from_numpy(g[nm]) for nm in g.files if nm.startswith(f"{k}_p_")}
        net.load_state_dict(sd, strict=True)
        net = net.to(dev)
        x = torch.from_numpy(g[f"{k}_x"]).to(dev).requires_grad_(True)
        y = net(x)
        assert y.shape == g[f"{k}_y"].shape
        assert _rel(y.detach().cpu().numpy(), g[f"{k}_y"]) < 1e-5, k
        w = torch.linspace(0.5, 1.5, y.numel()).view(y.shape).to(dev)
        (y * w).sum().backward()
        assert _rel(x.grad.cpu().numpy(), g[f"{k}_dx"]) < 2e-5, k
        for nm, p in net.named_parameters():
            assert _rel(p.grad.cpu().numpy(), g[f"{k}_g_{nm}"]) < 2e-5, (k, nm)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["tcn", "dstcn"])
def test_spectral_tcn_models_vs_oracle(dev, kind):
    """full models at the shipped geometry (513 bins x 345 frames, 5 blocks of 96 channels, 13 taps): forward 1e-5,
    gradients 2e-5 of each tensor's max (the log-spectrogram front end is the K4 kernel with an identity filter bank)."""
    from mod_extraction_amd import models
    from oracle import models as om
    torch.manual_seed(11)
    ref = (om.SpectralTCN if kind == "tcn" else om.SpectralDSTCN)(n_samples=88200)
    with torch.no_grad():
        for b in ref.tcn.blocks:
            b.act.weight.uniform_(0.05, 0.45)
    mine = (models.SpectralTCN if kind == "tcn" else models.SpectralDSTCN)(n_samples=88200)
    mine.load_state_dict(ref.state_dict(), strict=True)
    mine = mine.to(dev)
    t = torch.arange(88200) / 44100.0
    x = (0.4 * torch.sin(2 * np.pi * 220.0 * t).view(1, 1, -1) + 0.3 * (torch.rand(2, 1, 88200) * 2 - 1)).clamp(-1, 1)
    # the fp64 evaluation of the oracle arbitrates the gradients: this image's fp32 CPU conv1d returns a WRONG weight
    # gradient for one DSTCN shape (96 -> 96 channels, 87 frames, dilation 4, stride 2: relative error ~1 against fp64,
    # reproduced with a bare nn.Conv1d), so the fp32 oracle is only used for the forward value
    ref64 = (om.SpectralTCN if kind == "tcn" else om.SpectralDSTCN)(n_samples=88200).double()
    ref64.load_state_dict({k: v.double() for k, v in ref.state_dict().items()})
    out_r = ref(x)
    out_m = mine(x.to(dev))
    out_64 = ref64(x.double())
    assert out_m.shape == out_r.shape == ((2, 1, 345) if kind == "tcn" else (2, 2))
    assert float((out_m.detach().cpu() - out_r.detach()).abs().max()) < 1e-5 * max(1.0, float(out_r.detach().abs().max()))
    assert float((out_m.detach().cpu().double() - out_64.detach()).abs().max()) < 1e-5
    w = torch.linspace(0.5, 1.5, out_r.numel()).view_as(out_r)
    (out_64 * w.double()).sum().backward()
    (out_m * w.to(dev)).sum().backward()
    g64 = dict(ref64.named_parameters())
    for nm, p in mine.named_parameters():
        assert p.grad is not None, nm
        a, b = p.grad.cpu().double(), g64[nm].grad
        assert float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)) < 2e-5, nm


@pytest.mark.gpu
@pytest.mark.parametrize("kind,kw", [
    ("tcn", dict(n_samples=22272, out_channels=[48, 64, 96], dilations=[1, 4, 2], kernel_size=7, latent_dim=2, use_res=True)),
    ("tcn", dict(n_samples=30000, out_channels=[32, 32], dilations=[1, 8], kernel_size=3, use_ln=False, use_res=False)),
    ("dstcn", dict(n_samples=22272, out_channels=[24, 40, 56], dilations=[1, 2, 4], kernel_size=5)),
], ids=["tcn-3blocks-k7", "tcn-no-ln-no-res", "dstcn-3blocks-k5"])
def test_spectral_tcn_other_geometries(dev, kind, kw):
    """SpectralTCN / SpectralDSTCN away from configs/models/spectral_tcn.yml: other lengths, block counts, channel widths,
    dilation orders, kernel sizes, with and without LayerNorm / residual branches; forward 1e-5 against the fp32 oracle,
    gradients 2e-5 of each tensor's max against the fp64 oracle (see the test above for why fp64 arbitrates)."""
    from mod_extraction_amd import models
    from oracle import models as om
    torch.manual_seed(5)
    Ref = om.SpectralTCN if kind == "tcn" else om.SpectralDSTCN
    Mine = models.SpectralTCN if kind == "tcn" else models.SpectralDSTCN
    ref = Ref(**kw)
    with torch.no_grad():
        for b in ref.tcn.blocks:
            b.act.weight.uniform_(0.05, 0.45)
    mine = Mine(**kw)
    mine.load_state_dict(ref.state_dict(), strict=True)
    mine = mine.to(dev)
    n = kw["n_samples"]
    x = (0.3 * torch.sin(2 * np.pi * 330.0 * torch.arange(n) / 44100.0).view(1, 1, -1) + 0.4 * (torch.rand(3, 1, n) * 2 - 1)).clamp(-1, 1)
    ref64 = Ref(**kw).double()
    ref64.load_state_dict({k: v.double() for k, v in ref.state_dict().items()})
    out_r, out_m, out_64 = ref(x), mine(x.to(dev)), ref64(x.double())
    assert out_m.shape == out_r.shape
    assert float((out_m.detach().cpu() - out_r.detach()).abs().max()) < 1e-5 * max(1.0, float(out_r.detach().abs().max()))
    w = torch.linspace(0.5, 1.5, out_r.numel()).view_as(out_r)
    (out_64 * w.double()).sum().backward()
    (out_m * w.to(dev)).sum().backward()
    g64 = dict(ref64.named_parameters())
    for nm, p in mine.named_parameters():
        assert p.grad is not None, nm
        a, b = p.grad.cpu().double(), g64[nm].grad
        assert float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)) < 2e-5, nm
