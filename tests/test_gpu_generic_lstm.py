"""GPU parity of LSTMEffectModel OUTSIDE the shipped 1 / 1 / 64 / 1 size (mod_extraction/models.py:311-339 with other in_ch /
out_ch / n_hidden / latent_dim) and of the TBPTT step with a param_model (lightning.py:212,344-347,371-375): the general
recurrence of csrc/lstm_generic.hip + fp32 matrix-core GEMMs against torch's nn.LSTM / nn.Linear on the CPU (oracle/models.py's
LSTMEffectModel is the reference's module graph).  Tolerances as in tests/test_gpu_lstm.py: outputs 2e-5 absolute (tanh
range), gradients 1e-4 relative to each tensor's max magnitude.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F
from torch import nn

from oracle import models as om

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _pair(dev, in_ch, out_ch, n_hidden, latent_dim, seed=0):
    from mod_extraction_amd import models as am
    torch.manual_seed(seed)
    ref = om.LSTMEffectModel(in_ch, out_ch, n_hidden, latent_dim)
    mine = am.LSTMEffectModel(in_ch, out_ch, n_hidden, latent_dim)
    mine.load_state_dict(ref.state_dict(), strict=True)
    assert mine.generic
    return ref, mine.to(dev)


SIZES = [(1, 1, 32, 1), (1, 1, 64, 3), (2, 2, 48, 1), (1, 1, 100, 2), (1, 3, 16, 1), (1, 1, 256, 1), (1, 1, 7, 1)]


@pytest.mark.parametrize("in_ch,out_ch,n_hidden,latent_dim", SIZES)
def test_generic_lstm_two_chunks_forward_backward_vs_torch(dev, in_ch, out_ch, n_hidden, latent_dim):
    """Two consecutive calls (the second starts from the first's detached state, as a TBPTT step does): outputs, the carried
    state, every parameter gradient and d loss / d latent of the second call."""
    ref, mine = _pair(dev, in_ch, out_ch, n_hidden, latent_dim)
    B, Tn = 3, 300 if n_hidden < 200 else 90
    g = torch.Generator().manual_seed(5)
    ref.clear_hidden(); mine.clear_hidden()
    for call in range(2):
        x = torch.rand(B, in_ch, Tn, generator=g) * 1.6 - 0.8
        lat = torch.rand(B, latent_dim, Tn, generator=g)
        w = torch.rand(B, max(in_ch, out_ch), Tn, generator=g) - 0.5
        lat_r = lat.clone().requires_grad_(True)
        lat_m = lat.to(dev).requires_grad_(True)
        ref.zero_grad(); mine.zero_grad()
        y_r = ref(x, lat_r)
        (y_r * w).sum().backward()
        y_m = mine(x.to(dev), lat_m)
        (y_m * w.to(dev)).sum().backward()
        assert y_m.shape == y_r.shape
        assert float((y_m.detach().cpu() - y_r.detach()).abs().max()) < 2e-5
        for (name, p), q in zip(mine.named_parameters(), ref.parameters()):
            e = _rel(p.grad.cpu(), q.grad)
            assert e < 1e-4, (call, name, e)
        assert _rel(lat_m.grad.cpu(), lat_r.grad) < 1e-4
        for hm, hr in zip(mine.hidden, ref.hidden):
            assert float((hm.cpu().reshape(-1) - hr.detach().reshape(-1)).abs().max()) < 2e-5
        ref.detach_hidden(); mine.detach_hidden()


def test_generic_lstm_single_step_chunk(dev):
    ref, mine = _pair(dev, 1, 1, 24, 2)
    x, lat = torch.rand(2, 1, 1) - 0.5, torch.rand(2, 2, 1)
    y_r = ref(x, lat)
    y_r.sum().backward()
    y_m = mine(x.to(dev), lat.to(dev))
    y_m.sum().backward()
    assert float((y_m.detach().cpu() - y_r.detach()).abs().max()) < 1e-6
    for p, q in zip(mine.parameters(), ref.parameters()):
        assert _rel(p.grad.cpu(), q.grad) < 1e-5


def test_shipped_size_keeps_the_fused_kernels_and_rejects_unbroadcastable_channels(dev):
    from mod_extraction_amd import models as am
    assert not am.LSTMEffectModel(1, 1, 64, 1).generic
    with pytest.raises(ValueError):
        am.LSTMEffectModel(in_ch=2, out_ch=3, n_hidden=8, latent_dim=1)


class _ParamNet(nn.Module):
    """A stand-in param_model (the reference ships none; lightning.py:212 takes any nn.Module wet -> (B, P)): two statistics of
    the clip through a linear layer.  Plain torch ops: its gradient arrives through autograd from the effect model's latent."""

    def __init__(self, P: int) -> None:
        super().__init__()
        self.lin = nn.Linear(2, P)

    def forward(self, wet):
        feats = torch.stack([wet.abs().mean(dim=(1, 2)), (wet ** 2).mean(dim=(1, 2)).sqrt()], dim=1)
        return torch.tanh(self.lin(feats))


@pytest.mark.parametrize("n_hidden,P", [(64, 2), (20, 1)])
def test_tbptt_step_with_param_model_vs_torch_loop(dev, n_hidden, P):
    """TBPTTLFOEffectModeling with a param_model and ground-truth LFOs: three optimizer steps of one batch against the
    reference loop (lightning.py:339-384) restated on torch modules on the CPU -- warm-up without loss, then per chunk: param_model
    re-evaluated, latent = cat[LFO chunk, params repeated], LSTM, L1, backward into BOTH models, AdamW."""
    import copy
    from mod_extraction_amd import lightning, models as am, optim
    torch.manual_seed(3); np.random.seed(3)
    B, W, S, n = 3, 256, 512, 256 + 3 * 512 + 100
    ref_em = om.LSTMEffectModel(1, 1, n_hidden, 1 + P)
    ref_pm = _ParamNet(P)
    em = am.LSTMEffectModel(1, 1, n_hidden, 1 + P); em.load_state_dict(ref_em.state_dict())
    pm = copy.deepcopy(ref_pm)
    mod = lightning.TBPTTLFOEffectModeling(W, S, em, lfo_model=None, param_model=pm, model_smooth_n_frames=0, should_stretch=False,
                                           discard_invalid_lfos=False, loss_dict={"l1": 1.0, "esr": 0.0, "dc": 0.0}).to(dev).train()
    opt = optim.FlatAdamW([p for p in mod.parameters() if p.requires_grad], lr=1e-3, betas=(0.8, 0.99))
    ref_params = list(ref_em.parameters()) + list(ref_pm.parameters())
    assert opt.numel == sum(p.numel() for p in ref_params)
    ref_opt = torch.optim.AdamW(ref_params, lr=1e-3, betas=(0.8, 0.99))
    g = torch.Generator().manual_seed(9)
    dry = torch.rand(B, 1, n, generator=g) * 1.6 - 0.8
    wet = (0.7 * dry + 0.25 * torch.roll(dry, 3, -1)).clamp(-1, 1)
    t = torch.linspace(0, 1, 60)
    mod_sig = 0.5 + 0.5 * torch.cos(2 * np.pi * (1.0 + torch.arange(B).view(B, 1)) * t)
    loss = mod.training_step((dry.to(dev), wet.to(dev), mod_sig.to(dev), None), 0, optimizer=opt)
    assert opt.step_count == 3

    lfo = F.interpolate(mod_sig.unsqueeze(1), size=n, mode="linear", align_corners=True)
    ref_em.clear_hidden()
    with torch.no_grad():
        p0 = ref_pm(wet).unsqueeze(-1)
        chunks = [ref_em(dry[..., :W], torch.cat([lfo[..., :W], p0.repeat(1, 1, W)], dim=1))]
    ref_em.detach_hidden()
    for s in range(W, n - S + 1, S):
        ref_opt.zero_grad()
        p = ref_pm(wet).unsqueeze(-1)
        y = ref_em(dry[..., s:s + S], torch.cat([lfo[..., s:s + S], p.repeat(1, 1, S)], dim=1))
        F.l1_loss(y, wet[..., s:s + S]).backward()
        ref_opt.step()
        ref_em.detach_hidden()
        chunks.append(y.detach())
    wet_hat = torch.cat(chunks, dim=-1)
    m = wet_hat.size(-1)
    want = F.l1_loss(wet_hat[..., W:m], wet[..., W:m])
    assert abs(float(loss) - float(want)) < 2e-5 * max(1.0, abs(float(want))), (float(loss), float(want))
    for (name, p), q in zip(list(em.named_parameters()) + list(pm.named_parameters()), ref_params):
        d = float((p.detach().cpu() - q.detach()).abs().max())
        assert d < 2e-4, (name, d)                                  # three steps of lr 1e-3: a sign flip of one step would be 2e-3
    # validation: no optimizer, the same forward
    out = mod.validation_step((dry.to(dev), wet.to(dev), mod_sig.to(dev), None), 0)
    assert out[1]["wet_hat"].shape == (B, 1, 3 * S)


def test_tbptt_step_with_a_wider_effect_model_and_unfrozen_extractor(dev):
    """The general step composes with freeze_lfo_model = False: extractor (general CNN kernels) -> moving average -> resampling ->
    a 32-unit LSTM; both models move, the loss of the trained chunks falls over a few batches."""
    from mod_extraction_amd import lightning, models as am, optim
    torch.manual_seed(12)
    n, W, S = 22272, 1024, 4096
    cnn = am.Spectral2DCNN(in_ch=2, n_samples=n, n_mels=32, kernel_size=(3, 5), out_channels=[8, 8], temp_dilations=[1, 2],
                           pool_size=(2, 1))
    em = am.LSTMEffectModel(1, 1, 32, 1)
    mod = lightning.TBPTTLFOEffectModeling(W, S, em, lfo_model=cnn, freeze_lfo_model=False, should_stretch=False,
                                           discard_invalid_lfos=False, model_smooth_n_frames=4,
                                           loss_dict={"l1": 1.0, "esr": 0.5, "dc": 0.0}).to(dev).train()
    opt = optim.FlatAdamW([p for p in mod.parameters() if p.requires_grad], lr=2e-3, betas=(0.8, 0.99))
    g = torch.Generator().manual_seed(4)
    dry = (torch.rand(4, 1, n, generator=g) * 1.6 - 0.8).to(dev)
    wet = (0.5 * dry + 0.4 * torch.roll(dry, 5, -1)).clamp(-1, 1)
    before = [p.detach().clone() for p in list(cnn.parameters()) + list(em.parameters())]
    losses = [float(mod.training_step((dry, wet, None, None), 0, optimizer=opt)) for _ in range(4)]
    assert losses[-1] < losses[0], losses
    moved = [float((p.detach() - q).abs().max()) for p, q in zip(list(cnn.parameters()) + list(em.parameters()), before)]
    assert all(m > 0 for m in moved), moved
