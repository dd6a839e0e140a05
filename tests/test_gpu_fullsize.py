"""GPU: the hot path at BASELINE.json's FULL sizes (256 clips x 2 s @ 44.1 kHz per GPU, 256 mel bins, the
1.34 M-parameter CNN), checked through size-independent properties -- the CPU oracle needs minutes per step at this
size, so these tests do not call it:

* clips are independent units: any clip's result is bit-identical whether it is computed inside the full batch or in
  a small batch (the small batch also takes the non-XCD-remapped tile order, batch % 8 != 0);
* effects: mix = 0 returns the (clipped) dry signal exactly; the phaser with depth = 0 is a linear time-invariant
  filter, so scaling the input by a power of two scales the output exactly; the flanger is bit-identical between the
  in-kernel resampled LFO and the pre-resampled one;
* gradients are linear in the batch: dW(full batch) = dW(first half) + dW(second half) (1e-5: the f16x3 gradient
  scale and the slab partition differ between the runs, the arithmetic does not);
* a full-size train step is reproducible bit for bit and moves the loss downhill.
Tolerances are written where they are used; everything else is exact.
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SR, N, B = 44100, 88200, 256
CNN_CFG = dict(in_ch=2, n_samples=N, sr=SR, n_fft=1024, hop_len=256, n_mels=256, kernel_size=(5, 13),
               out_channels=[64] * 6, temp_dilations=[1, 1, 2, 4, 8, 16], pool_size=(2, 1), latent_dim=1,
               freq_mask_amount=0.25, time_mask_amount=0.25, use_ln=True)
LOSS = {"l1": 1.0, "fdl1": 5.0, "sdl1": 10.0, "mse": 0.0}


def _audio(dev, b, n, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return ((torch.rand((b, n), generator=g) * 2 - 1) * 0.89).to(dev)


@pytest.fixture(scope="module")
def model(dev):
    from mod_extraction_amd import models
    torch.manual_seed(7)
    m = models.Spectral2DCNN(**CNN_CFG).to(dev)
    m.eval()                                     # no SpecAugment: the mask is one draw per BATCH by design
    return m


def test_cnn_clips_are_independent_at_full_batch(dev, model):
    x = torch.stack([_audio(dev, B, N, 1), _audio(dev, B, N, 2)], dim=1)          # (B, 2, N) dry / wet
    with torch.no_grad():
        out_full, lat_full = model(x)
        pick = [0, 97, 255]                                                         # batch of 3: tile order not remapped
        out_sub, lat_sub = model(x[pick].contiguous())
    assert out_full.shape == (B, 1, N // 256 + 1) and lat_full.shape == (B, 64, N // 256 + 1)
    assert bool(torch.isfinite(out_full).all()) and float(out_full.min()) >= 0.0 and float(out_full.max()) <= 1.0
    assert torch.equal(out_full[pick], out_sub)
    assert torch.equal(lat_full[pick], lat_sub)


def test_weight_gradient_is_linear_in_the_batch(dev, model):
    x = torch.stack([_audio(dev, B, N, 3), _audio(dev, B, N, 4)], dim=1)
    g = torch.Generator(device="cpu").manual_seed(5)
    d_out = torch.randn((B, 1, N // 256 + 1), generator=g).to(dev) / B

    def grads(sl):
        model.zero_grad(set_to_none=True)
        out, _ = model(x[sl].contiguous())
        out.backward(d_out[sl].contiguous())
        return [p.grad.detach().clone() for p in model.parameters() if p.grad is not None]

    full = grads(slice(0, B))
    lo, hi = grads(slice(0, B // 2)), grads(slice(B // 2, B))
    assert len(full) == len(lo) == len(hi) and len(full) >= 20
    for gf, ga, gb in zip(full, lo, hi):
        scale = float(gf.abs().max()) + 1e-30
        assert float((gf - (ga + gb)).abs().max()) / scale < 1e-5          # fp32 parity tolerance of north_star
        assert bool(torch.isfinite(gf).all())


def test_flanger_full_size_properties(dev):
    from mod_extraction_amd import fx, modulations as amod
    x = _audio(dev, B, N, 11)
    rate = torch.exp(torch.empty(B).uniform_(math.log(0.5), math.log(3.0))).to(dev)
    phase = torch.empty(B).uniform_(0, 2 * math.pi).to(dev)
    lfo = amod.make_mod_signals(882, 441.0, rate, phase, None, None, None)                    # (B, 882)
    mod = fx.MonoFlangerChorusModule(B, 1, N, SR, max_min_delay_ms=1.0, max_lfo_delay_ms=10.0).to(dev)
    p = dict(feedback=torch.empty(B).uniform_(0, 0.7), min_delay_width=torch.rand(B), width=torch.empty(B).uniform_(0.25, 1),
             depth=torch.empty(B).uniform_(0.25, 1))
    # mix = 0: the wet path is multiplied by 0 -> exactly the clipped input (fx.py:116-118)
    y0 = mod(x.unsqueeze(1), lfo, mix=torch.zeros(B), **p)
    assert torch.equal(y0.squeeze(1), x.clamp(-1.0, 1.0))
    # in-kernel LFO resampling == resampling first (util.linear_interpolate_last_dim, align_corners=True)
    from mod_extraction_amd import util as autil
    mix = torch.empty(B).uniform_(0.25, 1)
    y_a = mod(x.unsqueeze(1), lfo, mix=mix, **p)
    y_b = mod(x.unsqueeze(1), autil.linear_interpolate_last_dim(lfo, N), mix=mix, **p)
    assert torch.equal(y_a, y_b)
    assert float(y_a.abs().max()) <= 1.0 and bool(torch.isfinite(y_a).all())


def test_phaser_full_size_properties(dev):
    from mod_extraction_amd import fx
    n_ph = 86                                                          # the phaser third of a 256-clip interwoven batch
    lead = torch.randint(14700, 88200, (n_ph,), dtype=torch.int32)
    src = _audio(dev, n_ph, N + 88200, 12) * 0.25
    prm = {"rate_hz": torch.exp(torch.empty(n_ph).uniform_(math.log(0.5), math.log(3.0))),
           "depth": torch.zeros(n_ph), "centre_frequency_hz": torch.exp(torch.empty(n_ph).uniform_(math.log(70.0), math.log(18000.0))),
           "feedback": torch.empty(n_ph).uniform_(0, 0.7), "mix": torch.empty(n_ph).uniform_(0.2, 1.0)}
    prm = {k: v.to(dev) for k, v in prm.items()}
    lead = lead.to(dev)
    # depth = 0: constant cut-off -> LTI; every operation is a multiply or an add, so a power-of-two gain commutes exactly
    y1 = fx.phaser_forward(src, prm, lead, SR, N)
    y2 = fx.phaser_forward(src * 0.5, prm, lead, SR, N)
    assert float(y1.abs().max()) < 1.0                                 # no clipping in play
    assert torch.equal(y2, y1 * 0.5)
    # mix = 0 returns the dry crop exactly
    dry = torch.empty((n_ph, N), device=dev)
    prm0 = dict(prm, mix=torch.zeros(n_ph, device=dev), depth=torch.full((n_ph,), 0.7, device=dev))
    y0 = fx.phaser_forward(src, prm0, lead, SR, N, dry_out=dry)
    idx = lead.long().unsqueeze(1) + torch.arange(N, device=dev).unsqueeze(0)
    assert torch.equal(dry, torch.gather(src, 1, idx))
    assert torch.equal(y0, dry.clamp(-1.0, 1.0))


def test_phaser_full_length_vs_oracle_and_bit_reference(dev):
    """The shipped (time-parallel scan) phaser at the HEADLINE geometry -- 2 s clips behind warm-ups of up to a whole LFO
    period (88 200 samples at 0.5 Hz), feedback up to 0.7, full modulation depth -- against the oracle
    (oracle_ref.c:orc_phaser, the JUCE restatement) over all 176 400 rendered samples, and against this package's own
    JUCE-order kernel.  1e-5 absolute on [-1, 1] audio (north_star); measured values are printed."""
    from mod_extraction_amd import fx
    from oracle import fx as ofx
    torch.manual_seed(31)
    leads = [88200, 0, 44100, 12345, 3, 70001, 88199, 1]
    n = len(leads)
    lead = torch.tensor(leads, dtype=torch.int32)
    src = torch.rand(n, N + 88200) * 1.6 - 0.8
    p = {"rate_hz": torch.tensor([0.5, 3.0, 1.0, 2.2, 0.9, 0.5, 0.61, 2.999]), "depth": torch.tensor([1.0, 1.0, 0.6, 0.9, 0.2, 1.0, 0.8, 1.0]),
         "centre_frequency_hz": torch.tensor([70.0, 18000.0, 440.0, 1300.0, 5000.0, 200.0, 9000.0, 100.0]),
         "feedback": torch.tensor([0.7, 0.7, 0.25, 0.5, 0.69, 0.7, 0.0, 0.7]), "mix": torch.tensor([1.0, 0.5, 0.5, 0.8, 1.0, 1.0, 1.0, 0.2])}
    pd = {k: v.to(dev) for k, v in p.items()}
    y = fx.phaser_forward(src.to(dev), pd, lead.to(dev), SR, N)
    y_exact = fx.phaser_forward(src.to(dev), pd, lead.to(dev), SR, N, exact_order=True)
    worst_o, worst_e = 0.0, 0.0
    for b in range(n):
        total = leads[b] + N
        ref = ofx.phaser_np(src[b:b + 1, :total].numpy(), p["rate_hz"][b:b + 1].numpy(), p["depth"][b:b + 1].numpy(),
                            p["centre_frequency_hz"][b:b + 1].numpy(), p["feedback"][b:b + 1].numpy(), p["mix"][b:b + 1].numpy(), SR)
        want = torch.from_numpy(ref[0, leads[b]:]).clamp(-1.0, 1.0)
        e_o = float((y[b].cpu() - want).abs().max())
        e_x = float((y_exact[b].cpu() - want).abs().max())
        worst_o, worst_e = max(worst_o, e_o), max(worst_e, e_x)
        assert e_o < 1e-5 and e_x < 1e-5, (b, e_o, e_x)
    d = float((y - y_exact).abs().max())
    print(f"[measured] phaser at 88200 + lead <= 88200, feedback <= 0.7: scan vs oracle {worst_o:.2e}, JUCE-order kernel vs "
          f"oracle {worst_e:.2e}, scan vs JUCE-order kernel {d:.2e} (gate 1e-5)")
    assert d < 1e-5


def test_full_size_train_step_is_reproducible_and_descends(dev):
    from mod_extraction_amd import data_modules, lightning, models, optim, trainer

    def run(n_steps):
        torch.manual_seed(43)
        np.random.seed(43)
        model = models.Spectral2DCNN(**CNN_CFG)
        module = lightning.LFOExtraction(model, sr=SR, use_dry=True, model_smooth_n_frames=0, should_stretch=False,
                                         loss_dict=LOSS).to(dev).train()
        opt = optim.FlatAdamW(module.parameters(), lr=1e-4, betas=(0.8, 0.99))
        batcher = data_modules.SyntheticFxBatcher(B, N, SR, ("flanger", "chorus", "phaser"), dev, audio_seed=43)
        batch = batcher.next_batch()
        runner = trainer.Trainer(log_fn=None)
        losses = [float(runner.train_step(module, opt, batch).detach()) for _ in range(n_steps)]   # the SAME batch every step
        flat = torch.cat([p.detach().flatten() for p in module.parameters()])
        return losses, flat

    la, pa = run(4)
    lb, pb = run(4)
    assert la == lb and torch.equal(pa, pb)              # bit-reproducible: no atomics on any data path
    assert all(math.isfinite(v) for v in la)
    assert la[-1] < la[0]                                # 4 AdamW steps on one batch reduce its loss
