"""GPU parity: K4 log-mel front end and the K5-K7 Spectral2DCNN stack (forward and every parameter
gradient) plus the K8 loss against the CPU oracle (oracle/models.py, torch fp32 on the host).

Tolerances (fp32, different summation orders on the two sides; north_star asks 1e-5 relative):
  * log-mel:  |diff| <= 1e-5 * |ref| + 2e-5 on the log scale (cells clipped at log(1e-7) excluded from rtol)
  * CNN forward (sigmoid LFO, latent): 1e-5 relative to the tensor's max magnitude
  * parameter gradients: 2e-5 relative to each gradient tensor's max magnitude
"""
import numpy as np
import pytest
import torch

from oracle import models as omodels

pytestmark = pytest.mark.gpu

CFG = dict(in_ch=2, n_fft=1024, hop_len=256, n_mels=256, kernel_size=(5, 13), out_channels=[64] * 6,
           temp_dilations=[1, 1, 2, 4, 8, 16], pool_size=(2, 1), latent_dim=1, freq_mask_amount=0.25,
           time_mask_amount=0.25, use_ln=True)


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def make_pair(dev, n_samples=88200, seed=0, **over):
    from mod_extraction_amd import models as amodels
    cfg = dict(CFG, n_samples=n_samples)
    cfg.update(over)
    torch.manual_seed(seed)
    ref = omodels.Spectral2DCNN(**cfg)
    with torch.no_grad():                       # PReLU slopes / biases away from their trivial init
        for m in ref.cnn:
            if isinstance(m, torch.nn.PReLU):
                m.weight.uniform_(0.05, 0.45)
    mine = amodels.Spectral2DCNN(**cfg)
    missing, unexpected = mine.load_state_dict(ref.state_dict(), strict=True), None
    return ref, mine.to(dev)


def audio(B, n, seed=1):
    g = torch.Generator().manual_seed(seed)
    dry = torch.rand(B, 1, n, generator=g) * 2 - 1
    t = torch.arange(n) / 44100.0
    tone = 0.4 * torch.sin(2 * np.pi * 220.0 * t * (1 + 0.3 * torch.rand(B, 1, 1, generator=g)))
    wet = 0.6 * dry + tone
    return torch.cat([0.5 * dry, wet.clamp(-1, 1)], dim=1)


def test_state_dict_keys_match_reference_layout(dev):
    ref, mine = make_pair(dev, n_samples=22272, n_mels=64)
    assert list(ref.state_dict().keys()) == list(mine.state_dict().keys())
    expect = {f"cnn.{i}.weight" for i in (1, 3, 5, 7, 9, 11, 13, 15, 17, 19, 21, 23)} | \
             {f"cnn.{i}.bias" for i in (1, 5, 9, 13, 17, 21)} | {"output.weight", "output.bias",
                                                                 "spectrogram.spectrogram.window",
                                                                 "spectrogram.mel_scale.fb"}
    assert set(mine.state_dict().keys()) == expect


@pytest.mark.parametrize("masks", [(0, 0, 0, 0), (10, 50, 100, 160)])
def test_logmel(dev, masks):
    ref, mine = make_pair(dev)
    x = audio(3, 88200)
    with torch.no_grad():
        want = ref.log_mel(x, masks)
        got = mine.log_mel(x.to(dev), masks).cpu()
    assert got.shape == (3, 2, 256, 352)
    assert torch.all(got[..., 345:] == 0)
    got = got[..., :345]
    floor = np.log(1e-7)
    live = want > floor + 1.0                      # cells well above the clip floor
    err = (got - want).abs()
    assert float(err[live].max()) <= 2e-5 + 1e-5 * float(want[live].abs().max()), float(err[live].max())
    assert float(err[~live].max()) < 0.05          # near the floor the log amplifies fp32 FFT noise
    assert torch.equal(got <= floor, want <= floor) or float(((got <= floor) != (want <= floor)).float().mean()) < 1e-3


def _loss(out):
    # smooth scalar that touches every output element with distinct weights
    w = torch.linspace(0.5, 1.5, out.numel(), device=out.device).view_as(out)
    return (out * w).sum() / out.numel()


oracle_forward_routed = omodels.forward_routed        # shared with __graft_entry__.smoke()


def run_pair(dev, ref, mine, x, masks):
    from mod_extraction_amd import models as amodels
    amodels.DEBUG_TAP = {}
    try:
        out_m, lat_m = mine(x.to(dev), masks)
        (_loss(out_m) + 0.1 * _loss(lat_m)).backward()
        tap = amodels.DEBUG_TAP
    finally:
        amodels.DEBUG_TAP = None
    out_r, lat_r, n_kinks = oracle_forward_routed(ref, x, masks, tap, mine.n_frames)
    (_loss(out_r) + 0.1 * _loss(lat_r)).backward()
    n_decisions = sum(v.numel() for k, v in tap.items() if k.startswith("amax"))
    assert n_kinks <= 64 + 1e-5 * n_decisions          # shared decisions are rare events (every one is tie-checked)
    assert rel_err(out_m.detach().cpu(), out_r.detach()) < 1e-5
    assert rel_err(lat_m.detach().cpu(), lat_r.detach()) < 1e-5
    gr = dict(ref.named_parameters())
    for name, p in mine.named_parameters():
        assert p.grad is not None, name
        e = rel_err(p.grad.cpu(), gr[name].grad)
        assert e < 2e-5, (name, e)


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
@pytest.mark.parametrize("n_samples,n_mels,B", [(22272, 64, 3), (88200, 256, 2), (20000, 64, 9)])
def test_cnn_forward_backward(dev, n_samples, n_mels, B, precision):
    """both conv arithmetic modes (split-fp16 matrix cores / exact fp32 MFMA) against the fp32 CPU oracle at
    the same 1e-5 (outputs) / 2e-5 (gradients) tolerances"""
    ref, mine = make_pair(dev, n_samples=n_samples, n_mels=n_mels)
    mine.conv_precision = precision
    ref.eval(); mine.eval()                        # no SpecAugment draw; masks injected explicitly
    run_pair(dev, ref, mine, audio(B, n_samples), (3, 11, 20, 41))


@pytest.mark.parametrize("knob", ["GPOOL_FUSED", "STATS_FUSED", "LN_FUSED"])
def test_cnn_forward_backward_with_a_fusion_switched_off(dev, knob, monkeypatch):
    """The unfused routes stay available as A/B knobs (MODEX_GPOOL=split: LayerNorm backward and pooled-operand pass as two
    kernels with the exact max|G| scale; MODEX_STATS=sweep: LayerNorm statistics by a sweep over the plane; MODEX_LN=sweep: the
    LayerNorm backward's own statistics sweep) and meet the same tolerances."""
    from mod_extraction_amd import models as amodels
    monkeypatch.setattr(amodels, knob, False)
    ref, mine = make_pair(dev, n_samples=22272, n_mels=64)
    ref.eval(); mine.eval()
    run_pair(dev, ref, mine, audio(3, 22272), (3, 11, 20, 41))


def test_cnn_single_channel_input(dev):
    ref, mine = make_pair(dev, n_samples=22272, n_mels=64, in_ch=1)
    ref.eval(); mine.eval()
    run_pair(dev, ref, mine, audio(2, 22272)[:, 1:2], (0, 0, 0, 0))


def test_lfo_loss_kernel(dev):
    from mod_extraction_amd import losses as alosses
    from oracle import losses as olosses
    torch.manual_seed(3)
    for n in (345, 342, 338):
        y_hat = torch.rand(7, n, requires_grad=True)
        y = torch.rand(7, n)
        w = {"l1": 1.0, "fdl1": 5.0, "sdl1": 10.0, "mse": 0.0}
        terms_r = {k: olosses.get_loss_func_by_name(k)(y_hat, y) for k in w}
        tot_r = sum(w[k] * terms_r[k] for k in w if w[k] > 0)
        tot_r.backward()
        yh = y_hat.detach().to(dev).requires_grad_(True)
        tot_m, terms_m = alosses.lfo_loss(yh, y.to(dev), w)
        tot_m.backward()
        assert abs(float(tot_m) - float(tot_r)) < 1e-6 * max(1.0, abs(float(tot_r)))
        for k in w:
            assert abs(float(terms_m[k]) - float(terms_r[k])) < 2e-6, k
        assert rel_err(yh.grad.cpu(), y_hat.grad) < 1e-5


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
@pytest.mark.parametrize("over", [
    dict(out_channels=[64] * 4, temp_dilations=[1, 16, 2, 8], n_mels=32, latent_dim=3),
    dict(out_channels=[64] * 3, temp_dilations=[2, 4, 1], n_mels=64, latent_dim=2),          # dilated FIRST block
    dict(out_channels=[64] * 5, temp_dilations=[1, 2, 4, 8, 16], n_mels=128, latent_dim=4),  # the class defaults' dilations
], ids=["4blocks-permuted", "3blocks-dilated-first", "5blocks-pow2"])
def test_cnn_other_members_of_the_supported_family(dev, over, precision):
    """Block counts, dilation orders, mel-bin counts and latent widths other than the shipped spectral_2dcnn.yml, inside
    what the f16x3 / fp32 block kernels cover (everything else takes the general kernels, test below)."""
    ref, mine = make_pair(dev, n_samples=22272, **over)
    mine.conv_precision = precision
    ref.eval(); mine.eval()
    run_pair(dev, ref, mine, audio(3, 22272), (2, 7, 10, 31))


def test_cnn_outside_the_family_takes_the_general_kernels(dev):
    """Configurations the f16x3 block kernels are not built for no longer raise: they run csrc/cnn_generic.hip
    (tests/test_gpu_generic_cnn.py holds their parity tests); the shipped family stays on the fast kernels."""
    from mod_extraction_amd import models as amodels
    assert not amodels.Spectral2DCNN(**{**CFG, "n_samples": 88200}).generic
    for other in (dict(kernel_size=(3, 3)), dict(pool_size=(3, 1), out_channels=[64] * 5, temp_dilations=[1, 2, 4, 8, 16]),
                  dict(out_channels=[32] * 6), dict(use_ln=False), dict(temp_dilations=[1, 1, 2, 4, 8, 32]), dict(n_mels=100),
                  dict(n_samples=200000)):
        assert amodels.Spectral2DCNN(**{**CFG, "n_samples": 88200, **other}).generic
    with pytest.raises(ValueError):             # 256 bins do not survive six poolings by 3
        amodels.Spectral2DCNN(**{**CFG, "n_samples": 88200, "pool_size": (3, 1)})
