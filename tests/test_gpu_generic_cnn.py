"""GPU parity of Spectral2DCNN OUTSIDE the shipped 5x13 / 64-channel / pool (2,1) family (mod_extraction/models.py:127-215 with
other kernel sizes, channel lists, dilations, pooling windows, use_ln=False, in_ch, latent_dim, frame counts -- the class's own
defaults among them): csrc/cnn_generic.hip + the fp32 matrix-core GEMM against the CPU oracle (oracle/models.py, whose
Spectral2DCNN is the reference's module graph built from torch.nn layers).

Tolerances (exact fp32 products on both sides, different summation orders): outputs 1e-5, parameter gradients 2e-5, each
relative to the tensor's max magnitude -- the gates of tests/test_gpu_cnn.py.
"""
import copy

import pytest
import torch

from oracle import models as omodels

pytestmark = pytest.mark.gpu


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def make_pair(dev, seed=0, **cfg):
    from mod_extraction_amd import models as amodels
    torch.manual_seed(seed)
    ref = omodels.Spectral2DCNN(**cfg)
    with torch.no_grad():
        for m in ref.cnn:
            if isinstance(m, torch.nn.PReLU):
                m.weight.uniform_(0.05, 0.45)
    mine = amodels.Spectral2DCNN(**cfg)
    mine.load_state_dict(ref.state_dict(), strict=True)
    assert mine.generic
    return ref, mine.to(dev)


def audio(B, C, n, seed=1):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, C, n, generator=g) * 2 - 1
    t = torch.arange(n) / 44100.0
    return (0.6 * x + 0.4 * torch.sin(2 * torch.pi * 330.0 * t * (1 + 0.3 * torch.rand(B, C, 1, generator=g)))).clamp(-1, 1)


def _loss(out):
    w = torch.linspace(0.5, 1.5, out.numel(), device=out.device).view_as(out)
    return (out * w).sum() / out.numel()


def run_pair(dev, ref, mine, x, masks=(0, 0, 0, 0), out_tol=1e-5, grad_tol=2e-5):
    """Gates against the fp32 oracle; a gradient that misses its gate is arbitrated in fp64 (a bias in front of a LayerNorm has
    a gradient that is a near-total cancellation: there the fp32 ORACLE is itself 1e-4 away from the fp64 value, and the device
    has to be no further from it than three times the oracle's own distance)."""
    out_m, lat_m = mine(x.to(dev), masks)
    (_loss(out_m) + 0.1 * _loss(lat_m)).backward()
    out_r, lat_r = ref(x, masks)
    (_loss(out_r) + 0.1 * _loss(lat_r)).backward()
    assert out_m.shape == out_r.shape and lat_m.shape == lat_r.shape
    assert rel_err(out_m.detach().cpu(), out_r.detach()) < out_tol
    assert rel_err(lat_m.detach().cpu(), lat_r.detach()) < out_tol
    gr = dict(ref.named_parameters())
    g64 = None
    for name, p in mine.named_parameters():
        assert p.grad is not None, name
        e = rel_err(p.grad.cpu(), gr[name].grad)
        if e >= grad_tol:
            if g64 is None:
                ref64 = copy.deepcopy(ref).double()
                ref64.zero_grad()
                o64, l64 = ref64(x.double(), masks)
                (_loss(o64) + 0.1 * _loss(l64)).backward()
                g64 = {k: v.grad for k, v in ref64.named_parameters()}
            e_mine, e_ref = rel_err(p.grad.cpu().double(), g64[name]), rel_err(gr[name].grad.double(), g64[name])
            assert e_mine < max(grad_tol, 3 * e_ref), (name, e, e_mine, e_ref)


CASES = {
    # the class's own defaults (models.py:129-145) at a quarter of the bins and a short clip: pool (3,1), five 64-channel
    # blocks, temp dilations 1..16, in_ch 1
    "class-defaults-small": dict(n_samples=22272, n_mels=243),
    # odd sizes everywhere: 3x5 kernels, growing channel list, bin dilations, three input channels, two latent dimensions
    "3x5-mixed-channels": dict(in_ch=3, n_samples=12000, n_mels=60, kernel_size=(3, 5), out_channels=[8, 12, 20],
                               bin_dilations=[1, 2, 1], temp_dilations=[1, 3, 5], pool_size=(3, 1), latent_dim=2),
    # even kernel sizes: Conv2d(padding="same") pads asymmetrically (aten: the extra row / column goes after)
    "even-kernels-no-ln": dict(in_ch=2, n_samples=9000, n_mels=50, kernel_size=(4, 6), out_channels=[16, 16],
                               bin_dilations=[1, 3], temp_dilations=[2, 1], pool_size=(2, 1), use_ln=False, latent_dim=1),
    # the shipped block shape with 32 channels and a bin count that does not divide (floor-mode pooling drops rows)
    "5x13-32ch-100mels": dict(in_ch=2, n_samples=22272, n_mels=100, kernel_size=(5, 13), out_channels=[32] * 4,
                              temp_dilations=[1, 2, 4, 32], pool_size=(2, 1), latent_dim=5),
    # pooling window 1 (no pooling) and a 1x1 kernel
    "pool1-1x1": dict(in_ch=1, n_samples=6000, n_mels=24, kernel_size=(1, 1), out_channels=[4, 6], pool_size=(1, 1)),
}


@pytest.mark.parametrize("name", list(CASES))
def test_generic_cnn_forward_backward_vs_oracle(dev, name):
    cfg = CASES[name]
    ref, mine = make_pair(dev, **cfg)
    ref.eval(); mine.eval()
    run_pair(dev, ref, mine, audio(3, cfg.get("in_ch", 1), cfg["n_samples"]))


def test_generic_cnn_with_specaugment_masks_and_state_dict_layout(dev):
    cfg = dict(in_ch=2, n_samples=12000, n_mels=48, kernel_size=(3, 7), out_channels=[8, 8], pool_size=(2, 1), use_ln=False,
               freq_mask_amount=0.25, time_mask_amount=0.25)
    ref, mine = make_pair(dev, **cfg)
    # without LayerNorm the reference's nn.Sequential is [Conv2d, MaxPool2d, PReLU] per block (models.py:184-191)
    assert list(ref.state_dict().keys()) == list(mine.state_dict().keys())
    assert {"cnn.0.weight", "cnn.0.bias", "cnn.2.weight", "cnn.3.weight", "cnn.5.weight"} <= set(mine.state_dict().keys())
    run_pair(dev, ref, mine, audio(2, 2, 12000), masks=(5, 14, 8, 19))
    mine.train()
    torch.manual_seed(5)
    f0, f1, t0, t1 = mine.draw_masks()                 # the training-mode draw stays the front end's
    assert 0 <= f0 <= f1 <= 48 and 0 <= t0 <= t1 <= mine.n_frames


def test_generic_cnn_long_clip_beyond_the_352_frame_pitch(dev):
    """200 000 samples = 782 frames: outside the fast kernels' 352-column planes, otherwise the shipped block shape"""
    cfg = dict(in_ch=2, n_samples=200000, n_mels=32, kernel_size=(5, 13), out_channels=[64] * 2, temp_dilations=[1, 16],
               pool_size=(2, 1))
    ref, mine = make_pair(dev, **cfg)
    ref.eval(); mine.eval()
    assert mine.n_frames == 782
    run_pair(dev, ref, mine, audio(2, 2, 200000))


def test_generic_cnn_im2col_chunking_gives_the_same_gradients(dev, monkeypatch):
    """The im2col matrix is built for a chunk of clips at a time; one clip per chunk must give what one chunk gives."""
    from mod_extraction_amd import cnn_generic
    cfg = CASES["3x5-mixed-channels"]
    ref, mine = make_pair(dev, **cfg)
    ref.eval(); mine.eval()
    x = audio(5, 3, cfg["n_samples"]).to(dev)
    outs = []
    for col_bytes in (1 << 30, 1):
        monkeypatch.setattr(cnn_generic, "COL_BYTES", col_bytes)
        mine.zero_grad()
        out, lat = mine(x)
        (_loss(out) + 0.1 * _loss(lat)).backward()
        outs.append([out.detach().clone()] + [p.grad.clone() for p in mine.parameters()])
    for a, b in zip(*outs):
        assert rel_err(a, b) < 2e-6            # (the weight gradient's clip partials are reduced in fp64 either way)


def test_generic_cnn_trains_through_lfo_extraction(dev):
    """LFOExtraction.training_step (lightning.py:96-158) on a class-default-style extractor: the loss falls over a few AdamW steps."""
    from mod_extraction_amd import lightning as alightning, models as amodels, optim
    torch.manual_seed(0)
    n = 22272
    model = amodels.Spectral2DCNN(in_ch=1, n_samples=n, n_mels=81, out_channels=[16] * 3, temp_dilations=[1, 2, 4])
    module = alightning.LFOExtraction(model, sr=44100, use_dry=False, model_smooth_n_frames=0, loss_dict={"l1": 1.0, "fdl1": 5.0, "sdl1": 10.0}).to(dev)
    opt = optim.FlatAdamW(module.parameters(), lr=2e-3, betas=(0.8, 0.99))
    g = torch.Generator().manual_seed(3)
    wet = (torch.rand(4, 1, n, generator=g) * 2 - 1).to(dev)
    t = torch.linspace(0, 1, 882)
    mod_sig = (0.5 + 0.5 * torch.cos(2 * torch.pi * (1.0 + torch.arange(4).view(4, 1)) * t)).to(dev)
    losses = []
    for _ in range(6):
        opt.zero_grad()
        loss = module.training_step((None, wet, mod_sig, None), 0)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert losses[-1] < losses[0], losses
