"""GPU: the UNFROZEN LFO model inside the TBPTT step (mod_extraction/lightning.py:258,344-366 with freeze_lfo_model: false):
the extractor is re-run in every optimizer step and trained through the effect model.  Pieces first (d loss / d lfo from the
BPTT kernel, the transposes of the resampling and of the moving average, against torch autograd), then the whole step
against a torch restatement of the reference loop on the CPU oracle's modules.  Tolerances as in test_gpu_lstm.py (LSTM
gradients 1e-4 norm-wise) and test_gpu_cnn.py (CNN gradients 2e-5 with the device's pooling / PReLU decisions shared)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import models as om

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("T,fused", [(1024, True), (777, True), (1000, False)])
def test_bptt_dlfo_vs_autograd(dev, T, fused):
    """mx_lstm_bwd_dgate + mx_lstm_dlfo: d loss / d latent of every sample against autograd through nn.LSTM (the latent is
    input column 0, models.py:328); the parameter gradients must be the bits of the kernels without the extra output."""
    from mod_extraction_amd import effect_losses, models as am
    torch.manual_seed(21)
    B = 3
    x = torch.rand(B, 1, T) * 1.6 - 0.8
    lat = torch.rand(B, 1, T, requires_grad=True)
    wet = (0.6 * x + 0.3 * torch.roll(x, 2, -1)).clamp(-1, 1)
    sd = om.LSTMEffectModel(1, 1, 64, 1).state_dict()
    ref = om.LSTMEffectModel(1, 1, 64, 1); ref.load_state_dict(sd)
    mine = am.LSTMEffectModel(1, 1, 64, 1); mine.load_state_dict(sd); mine = mine.to(dev)
    ref.clear_hidden()
    y_r = ref(x, lat)
    weights = {"l1": 1.0} if fused else {"l1": 0.3, "mse": 0.7, "esr": 0.4}
    from oracle import losses as olosses
    sum(w * olosses.get_loss_func_by_name(k)(y_r, wet) for k, w in weights.items()).backward()
    xd, ld, wd = x.to(dev), lat.detach().to(dev), wet.to(dev)
    mine.clear_hidden()
    stash = torch.empty((B, T, 384), device=dev)
    y_m, h0, c0 = mine.run_chunk(xd, ld, stash)
    g_a, g_b = torch.empty(am.LSTM_NPARAM, device=dev), torch.empty(am.LSTM_NPARAM, device=dev)
    if fused:
        dlat = mine.bptt_chunk_dlfo(xd, ld, y_m, stash, h0, c0, g_a, wet=wd, loss_scale=1.0 / (B * T))
        mine.bptt_l1_chunk(xd, ld, y_m, wd, stash, h0, c0, 1.0 / (B * T), g_b)
    else:
        dy = effect_losses.effect_loss_grad(y_m, wd.contiguous(), weights)
        dlat = mine.bptt_chunk_dlfo(xd, ld, y_m, stash, h0, c0, g_a, dy=dy)
        mine.bptt_chunk(xd, ld, y_m, dy, stash, h0, c0, g_b)
    assert torch.equal(g_a, g_b)
    e = _rel(dlat.cpu(), lat.grad)
    print(f"[measured] d loss / d latent, T={T}: rel err {e:.2e}")
    assert e < 1e-4, e


@pytest.mark.parametrize("n_in,n_out,j0,j_len", [(81, 20500, 1024, 6000), (345, 86410, 85386, 1024), (338, 86410, 0, 86410),
                                                 (88, 88, 10, 20), (7, 1000, 990, 10), (882, 345, 0, 345)])
def test_interp_transpose_vs_autograd(dev, n_in, n_out, j0, j_len):
    """mx_interp_linear_bwd against autograd of F.interpolate(mode='linear', align_corners=True) (util.py:15-29) with the
    upstream gradient confined to a window of the output axis."""
    from mod_extraction_amd import util as autil
    torch.manual_seed(5)
    x = torch.rand(3, n_in, requires_grad=True)
    y = F.interpolate(x.unsqueeze(1), size=n_out, mode="linear", align_corners=True).squeeze(1) if n_in != n_out else x * 1.0
    g = torch.zeros(3, n_out)
    g[:, j0:j0 + j_len] = torch.randn(3, j_len)
    y.backward(g)
    dx = autil.linear_interpolate_last_dim_bwd(g[:, j0:j0 + j_len].contiguous().to(dev), n_in, n_out, j0)
    e = _rel(dx.cpu(), x.grad)
    assert e < 1e-5, e


def test_unfrozen_lfo_tbptt_step_vs_oracle(dev):
    """TBPTTLFOEffectModeling with freeze_lfo_model = False: two batches of ONE optimizer step each (so that the decisions the
    device's CNN backward records belong to that step) against the reference loop restated in torch on the oracle's modules:
    extractor forward WITH graph -> unfold / mean (lightning.py:288-289) -> linear_interpolate_last_dim -> the step's chunk ->
    LSTM -> L1 -> backward into BOTH models -> AdamW.  Checked: the step loss, every gradient of the first step (LSTM 1e-4,
    CNN 1e-4 norm-wise with shared pooling / PReLU decisions), the parameters after each step."""
    from mod_extraction_amd import lightning, models as am, optim
    n, sr, B, W, S, k = 22272, 44100, 3, 1024, 12000, 8
    cfg = dict(in_ch=2, n_samples=n, sr=sr, n_fft=1024, hop_len=256, n_mels=64, kernel_size=(5, 13), out_channels=[64] * 6,
               temp_dilations=[1, 1, 2, 4, 8, 16], pool_size=(2, 1), latent_dim=1, freq_mask_amount=0.0, time_mask_amount=0.0,
               use_ln=True)
    torch.manual_seed(31); np.random.seed(31)
    ref_cnn = om.Spectral2DCNN(**cfg).train()
    ref_em = om.LSTMEffectModel(1, 1, 64, 1)
    cnn = am.Spectral2DCNN(**cfg); cnn.load_state_dict(ref_cnn.state_dict())
    em = am.LSTMEffectModel(1, 1, 64, 1); em.load_state_dict(ref_em.state_dict())
    mod = lightning.TBPTTLFOEffectModeling(W, S, em, lfo_model=cnn, freeze_lfo_model=False, should_stretch=False,
                                           discard_invalid_lfos=False, model_smooth_n_frames=k,
                                           loss_dict={"l1": 1.0, "esr": 0.0, "dc": 0.0}).to(dev).train()
    assert all(p.requires_grad for p in cnn.parameters()) and cnn.training
    params = [p for p in mod.parameters() if p.requires_grad]
    opt = optim.FlatAdamW(params, lr=1e-4, betas=(0.8, 0.99))
    assert opt.numel == am.LSTM_NPARAM + sum(p.numel() for p in ref_cnn.parameters())
    ref_opt = torch.optim.AdamW(list(ref_em.parameters()) + list(ref_cnn.parameters()), lr=1e-4, betas=(0.8, 0.99))
    F_frames = n // 256 + 1
    n_c = int(((F_frames - k + 1) / F_frames) * n)                     # lightning.py:321
    assert (n_c - W) // S == 1
    lo = (n - n_c) // 2
    g = torch.Generator().manual_seed(77)
    kinks = {}                                               # width of the decisions shared with the device (tie rule 1e-4 here)
    for step in range(2):
        dry = torch.rand(B, 1, n, generator=g) * 1.6 - 0.8
        wet = (0.7 * dry + 0.25 * torch.roll(dry, 3, -1)).clamp(-1, 1)
        am.DEBUG_TAP = {}
        try:
            loss = mod.training_step((dry.to(dev), wet.to(dev), None, None), 0, optimizer=opt)
            tap = am.DEBUG_TAP
        finally:
            am.DEBUG_TAP = None
        grads_m = opt.flat_grad.clone().cpu()

        # ---- the reference loop (lightning.py:310-384) on the oracle's modules, the device's decisions shared
        x_in = torch.cat([dry, wet], dim=1)
        def extract():
            hat, _, _ = om.forward_routed(ref_cnn, x_in, (0, 0, 0, 0), tap, cnn.n_frames, tie_tol=1e-4, kink_stats=kinks)
            hs = hat.squeeze(1).unfold(-1, k, 1).mean(-1)
            return F.interpolate(hs.unsqueeze(1), size=n_c, mode="linear", align_corners=True)
        dry_c, wet_c = dry[..., lo:lo + n_c], wet[..., lo:lo + n_c]
        ref_em.clear_hidden()
        with torch.no_grad():
            lfo0 = extract()
            ref_em(dry_c[..., :W], lfo0[..., :W])
        ref_em.detach_hidden()
        ref_opt.zero_grad()
        lfo1 = extract()
        y = ref_em(dry_c[..., W:W + S], lfo1[..., W:W + S])
        l_r = F.l1_loss(y, wet_c[..., W:W + S])
        l_r.backward()
        if step == 0:
            off = 0
            for name, p in list(ref_em.named_parameters()) + list(ref_cnn.named_parameters()):
                kk = p.numel()
                e = _rel(grads_m[off:off + kk], p.grad.reshape(-1))
                # LSTM parameters: the recurrence's 1e-4; CNN parameters (behind it in the flat buffer): the suite's 2e-5 CNN gate
                # and then some for the chain through the LSTM's d loss / d lfo (measured 8.2e-6)
                assert e < (1e-4 if off < 17473 else 3e-5), (name, e)
                off += kk
            assert off == opt.numel
        ref_opt.step()
        ref_em.detach_hidden()
        # the step object logs the loss of the whole concatenation (here: the one trained chunk)
        assert abs(float(loss) - float(l_r)) < 1e-5 * max(1.0, abs(float(l_r))), (step, float(loss), float(l_r))
        for (name, p), q in zip(list(em.named_parameters()) + list(cnn.named_parameters()),
                                list(ref_em.parameters()) + list(ref_cnn.parameters())):
            d = float((p.detach().cpu() - q.detach()).abs().max())
            assert d < 2.5e-4 * (step + 1), (step, name, d)           # <= 2 lr per step (Adam's first steps are ~ lr sign(g))
    print(f"unfrozen step: decisions shared with the device {kinks.get('n', 0)}, of them {kinks.get('n_wide', 0)} with |delta| between "
          f"2e-6 and 1e-4 of the tensor's max (widest {kinks.get('max_rel', 0.0):.2e})")
    assert kinks.get("max_rel", 0.0) <= 1e-4                   # widest decision shared with the device (measured_errors.json)
    assert kinks.get("n_wide", 0) / max(1, kinks.get("n", 0)) <= 1e-0
    assert opt.step_count == 2


@pytest.mark.parametrize("smooth", [0, 5])
@pytest.mark.parametrize("max_n_corners", [16, 2])
def test_stretch_corners_gradient_vs_autograd(dev, smooth, max_n_corners):
    """mx_stretch_corners_bwd against torch autograd through the reference's operation sequence restated out of place (oracle
    stretch_corners_torch, modulations.py:260-307, with a tensor-valued first anchor and last target; the reference's in-place
    original raises under autograd, see there): LFO-like rows with 1-7 corners, rows above max_n_corners (identity), a monotone
    row (one segment from sample 0 to the last sample).  Values of the restatement are checked against the bit-exact kernel first."""
    from mod_extraction_amd import modulations as amod
    from oracle import modulations as omod
    torch.manual_seed(8)
    n = 90
    t = torch.arange(n) / n
    rows = [0.5 + 0.4 * torch.sin(2 * np.pi * (f * t + ph)) * (0.6 + 0.4 * t) for f, ph in
            [(1.0, 0.1), (2.3, 0.4), (0.6, 0.8), (3.4, 0.0), (1.7, 0.55), (0.2, 0.3)]]
    rows.append(0.1 + 0.8 * t ** 2)                                  # monotone: no corner
    x = torch.stack(rows).clamp(0, 1).float()                       # (smooth rows: noise would add a corner at every other sample)
    xr = x.clone().requires_grad_(True)
    y_r = omod.stretch_corners_torch(xr, max_n_corners, smooth)
    y_m = amod.stretch_corners(x.to(dev), max_n_corners, smooth)
    assert float((y_m.cpu() - y_r.detach()).abs().max()) < 1e-6
    g = torch.randn(y_r.shape)
    y_r.backward(g)
    dx = amod.stretch_corners_bwd(x.to(dev), g.to(dev), max_n_corners, smooth)
    e = _rel(dx.cpu(), xr.grad)
    print(f"[measured] stretch_corners gradient (smooth {smooth}, max corners {max_n_corners}): rel err {e:.2e}")
    assert e < 1e-4, e                                              # (autograd's route subtracts the segment minimum twice: its cancellation noise)
    n_stretched = int(((y_r.detach() - (x if smooth <= 1 else x.unfold(-1, smooth, 1).mean(-1))).abs().amax(-1) > 1e-3).sum())
    assert n_stretched >= (5 if max_n_corners == 16 else 2)


def test_unfrozen_lfo_with_stretch_trains_both_models(dev):
    """The class defaults (should_stretch = True) with an unfrozen extractor: one step runs, every parameter of both models
    receives a finite gradient, and the extractor's gradient equals -- through the chain resampling^T -> stretch^T ->
    moving-average^T evaluated by torch autograd on the CPU from the SAME d loss / d lfo -- what the CNN was handed."""
    from mod_extraction_amd import lightning, models as am, optim
    n, B, W, S, k = 22272, 3, 1024, 12000, 8
    cfg = dict(in_ch=2, n_samples=n, sr=44100, n_fft=1024, hop_len=256, n_mels=64, kernel_size=(5, 13), out_channels=[64] * 6,
               temp_dilations=[1, 1, 2, 4, 8, 16], pool_size=(2, 1), latent_dim=1, use_ln=True)
    torch.manual_seed(41)
    cnn, em = am.Spectral2DCNN(**cfg), am.LSTMEffectModel()
    mod = lightning.TBPTTLFOEffectModeling(W, S, em, lfo_model=cnn, freeze_lfo_model=False, discard_invalid_lfos=False,
                                           model_smooth_n_frames=k).to(dev).train()
    assert mod.should_stretch
    opt = optim.FlatAdamW([p for p in mod.parameters() if p.requires_grad], lr=1e-4, betas=(0.8, 0.99))
    before = opt.flat_param.clone()
    g = torch.Generator().manual_seed(3)
    dry = torch.rand(B, 1, n, generator=g) * 1.6 - 0.8
    wet = (0.7 * dry + 0.25 * torch.roll(dry, 3, -1)).clamp(-1, 1)
    loss = mod.training_step((dry.to(dev), wet.to(dev), None, None), 0, optimizer=opt)
    assert loss is not None and torch.isfinite(loss)
    assert torch.isfinite(opt.flat_grad).all()
    n_lstm = am.LSTM_NPARAM
    assert float(opt.flat_grad[:n_lstm].abs().max()) > 0 and float(opt.flat_grad[n_lstm:].abs().max()) > 0
    assert float((opt.flat_param - before)[n_lstm:].abs().max()) > 0          # the extractor moved


def test_unfrozen_lfo_needs_every_clip(dev):
    """The reference re-extracts the LFO of every clip inside the step without re-applying its validity filter
    (lightning.py:344-349): with clips dropped its shapes no longer match; here that is a ValueError."""
    from mod_extraction_amd import lightning, models as am, optim
    n = 22272
    cfg = dict(in_ch=2, n_samples=n, sr=44100, n_fft=1024, hop_len=256, n_mels=64, kernel_size=(5, 13), out_channels=[64] * 6,
               temp_dilations=[1, 1, 2, 4, 8, 16], pool_size=(2, 1), latent_dim=1, use_ln=True)

    class HalfValid(torch.nn.Module):                                # an extractor whose first row is a clean triangle, the rest flat
        def __init__(self, net):
            super().__init__()
            self.net, self.n_frames = net, net.n_frames
        def forward(self, x):
            hat, lat = self.net(x)
            tri = (1 - (torch.arange(hat.size(-1), device=hat.device) / 20 % 2 - 1).abs()).view(1, 1, -1)
            keep = torch.zeros(hat.size(0), 1, 1, device=hat.device); keep[0] = 1
            return keep * tri + (1 - keep) * 0.5 + 0.0 * hat, lat
    torch.manual_seed(2)
    mod = lightning.TBPTTLFOEffectModeling(1024, 12000, am.LSTMEffectModel(), lfo_model=HalfValid(am.Spectral2DCNN(**cfg)),
                                           freeze_lfo_model=False, discard_invalid_lfos=True).to(dev).train()
    opt = optim.FlatAdamW([p for p in mod.parameters() if p.requires_grad], lr=1e-4)
    dry = torch.rand(3, 1, n, device=dev) - 0.5
    with pytest.raises(ValueError):
        mod.training_step((dry, dry.clone(), None, None), 0, optimizer=opt)
