"""GPU parity: K10 LSTM-64 effect model (forward with the 7 shipped weight files, BPTT gradients of a
1024-sample chunk, the TBPTT step logic) and the effect-model losses, against the golden vectors of
the reference's own classes and the CPU oracle.  Tolerances: audio in [-1,1], 1e-5 absolute;
gradients 1e-4 relative to each tensor's max (fp32, ~1k-step recurrences, device tanh/exp)."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import lightning as ol, losses as olosses, models as om

pytestmark = pytest.mark.gpu


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_lstm_forward_with_shipped_weights(golden_dir, dev):
    from mod_extraction_amd import models as am
    g = load(golden_dir, "lstm.npz")
    x, lat = torch.from_numpy(g["x"]).to(dev), torch.from_numpy(g["latent"]).to(dev)
    T = x.size(-1) // 2
    for i in range(int(g["n_files"])):
        m = am.LSTMEffectModel(1, 1, 64, 1)
        sd = {k[len(f"w_{i}_"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(f"w_{i}_")}
        m.load_state_dict(sd, strict=True)            # the shipped state-dict keys load unchanged
        m = m.to(dev)
        m.clear_hidden()
        y1 = m(x[..., :T].contiguous(), lat[..., :T].contiguous())
        m.detach_hidden()
        y2 = m(x[..., T:], lat[..., T:])              # strided chunk views are accepted
        y = torch.cat([y1, y2], -1).cpu().numpy()
        assert np.abs(y - g[f"y_{i}"]).max() < 1e-5, str(g[f"name_{i}"])
        assert np.abs(m.hidden[0].cpu().numpy() - g[f"h_{i}"]).max() < 1e-5
        assert np.abs(m.hidden[1].cpu().numpy() - g[f"c_{i}"]).max() < 2e-5


def test_lstm_bptt_chunk_gradients(golden_dir, dev):
    from mod_extraction_amd import models as am
    g = load(golden_dir, "lstm.npz")
    torch.manual_seed(4)
    B, T = 3, 1024
    x = torch.rand(B, 1, 2 * T) * 1.6 - 0.8
    lat = torch.rand(B, 1, 2 * T)
    wet = (0.6 * x + 0.3 * torch.roll(x, 2, -1)).clamp(-1, 1)
    for i in (0, 3):
        sd = {k[len(f"w_{i}_"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(f"w_{i}_")}
        ref = om.LSTMEffectModel(1, 1, 64, 1); ref.load_state_dict(sd)
        mine = am.LSTMEffectModel(1, 1, 64, 1); mine.load_state_dict(sd); mine = mine.to(dev)
        # warm-up chunk, detach, then the chunk whose gradients are compared
        ref.clear_hidden(); ref(x[..., :T], lat[..., :T]); ref.detach_hidden()
        y_r = ref(x[..., T:], lat[..., T:])
        torch.nn.functional.l1_loss(y_r, wet[..., T:]).backward()
        xd, ld, wd = x.to(dev), lat.to(dev), wet.to(dev)
        mine.clear_hidden(); mine.run_chunk(xd[..., :T], ld[..., :T]); mine.detach_hidden()
        stash = torch.empty((B, T, 384), device=dev)
        y_m, h0, c0 = mine.run_chunk(xd[..., T:], ld[..., T:], stash)
        grad = torch.empty(am.LSTM_NPARAM, device=dev)
        mine.bptt_l1_chunk(xd[..., T:], ld[..., T:], y_m, wd[..., T:], stash, h0, c0, 1.0 / (B * T), grad)
        assert float((y_m.cpu() - y_r.detach()).abs().max()) < 1e-5
        flat_r = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
        assert [n for n, _ in ref.named_parameters()] == ["lstm.weight_ih_l0", "lstm.weight_hh_l0", "lstm.bias_ih_l0",
                                                          "lstm.bias_hh_l0", "fc.weight", "fc.bias"]
        off = 0
        for n, p in ref.named_parameters():
            k = p.numel()
            a, r = grad[off:off + k].cpu(), p.grad.reshape(-1)
            e = float((a - r).abs().max() / r.abs().max())
            assert e < 1e-4, (n, e)
            off += k
        assert off == am.LSTM_NPARAM == flat_r.numel()


def test_effect_losses(golden_dir, dev):
    from mod_extraction_amd import losses as alosses
    g = load(golden_dir, "losses.npz")
    wa, wb = torch.from_numpy(g["wa"]).to(dev), torch.from_numpy(g["wb"]).to(dev)
    for name in ("esr", "dc"):
        v = float(alosses.get_loss_func_by_name(name)(wa, wb))
        assert abs(v - float(g["w_" + name])) <= 1e-5 * abs(float(g["w_" + name])) + 1e-9, name
    from mod_extraction_amd.effect_losses import effect_loss_terms
    assert abs(float(effect_loss_terms(wa, wb)["l1"]) - float(g["w_l1"])) < 1e-6


def test_tbptt_training_step_vs_reference_golden(golden_dir, dev):
    """TBPTTLFOEffectModeling.common_step with ground-truth LFOs against the golden captured from the
    reference's own class: kept clips, crop length, number of optimizer steps, processed LFOs (bit-exact),
    wet_hat and the logged batch losses."""
    from mod_extraction_amd import lightning as al, models as am, optim
    g = load(golden_dir, "steps.npz")
    em = am.LSTMEffectModel(1, 1, 64, 1)
    em.load_state_dict({k[len("tb_init_"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("tb_init_")})
    mod = al.TBPTTLFOEffectModeling(256, 256, em, lfo_model=None, model_smooth_n_frames=8, should_stretch=True,
                                    max_n_corners=16, stretch_smooth_n_frames=0, discard_invalid_lfos=True,
                                    loss_dict={"l1": 1.0, "esr": 0.0, "dc": 0.0}).to(dev).train()
    opt = optim.FlatAdamW(mod.parameters(), lr=1e-4, betas=(0.8, 0.99))
    assert opt.numel == am.LSTM_NPARAM
    batch = (torch.from_numpy(g["tb_dry"]).to(dev), torch.from_numpy(g["tb_wet"]).to(dev),
             torch.from_numpy(g["tb_lfo"]).to(dev), None)
    loss, dd, _ = mod.common_step(batch, is_training=True, optimizer=opt, world_size=1)
    assert dd["dry"].shape[0] == int(g["tb_kept"])
    assert opt.step_count == int(g["tb_steps"])
    assert np.array_equal(dd["mod_sig_hat"].cpu().numpy(), g["tb_mod_sig_hat"])       # K9: bit-exact
    assert dd["wet_hat"].shape == g["tb_wet_hat"].shape
    assert np.abs(dd["wet_hat"].cpu().numpy() - g["tb_wet_hat"]).max() < 1e-4
    assert abs(float(loss) - float(g["tb_loss"])) < 1e-5
    names = [str(n) for n in g["tb_names"]]
    for n, v in zip(names, g["tb_logged"]):
        assert abs(float(mod.logged[n][-1]) - float(v)) < 1e-5 * max(1.0, abs(float(v))), n
    # weights after the Adam steps: every weight moved by O(lr) per step; weights whose gradient is ~eps
    # are ill-conditioned under Adam, so the bulk statistic is compared, not the max
    for k, v in em.state_dict().items():
        d = np.abs(v.cpu().numpy() - g[f"tb_final_{k}"])
        moved = np.abs(g[f"tb_final_{k}"] - g[f"tb_init_{k}"])
        assert np.median(d) < 0.02 * max(np.median(moved), 1e-9), k


def test_tbptt_full_length_vs_oracle(dev):
    """2 s clips, warm-up 1024 + 83 chunks of 1024 (configs/train_em_dry_wet.yml geometry), validation mode
    (no optimizer), frozen random-init CNN replaced by ground-truth LFOs."""
    from mod_extraction_amd import lightning as al, models as am
    from oracle import modulations as omod
    torch.manual_seed(6)
    B, n = 3, 88200
    dry = torch.rand(B, 1, n) * 1.6 - 0.8
    wet = (0.7 * dry + 0.2 * torch.roll(dry, 5, -1)).clamp(-1, 1)
    lfo = torch.stack([omod.make_mod_signal(345, 172.5, f, p, "cos") for f, p in ((1.0, 0.2), (2.2, 1.0), (1.5, 3.0))])
    sd = om.LSTMEffectModel().state_dict()
    ref = om.LSTMEffectModel(); ref.load_state_dict(sd)
    em = am.LSTMEffectModel(); em.load_state_dict(sd)
    mod = al.TBPTTLFOEffectModeling(1024, 1024, em, lfo_model=None, loss_dict={"l1": 1.0, "esr": 0.0, "dc": 0.0}).to(dev).eval()
    loss, dd, _ = mod.validation_step((dry.to(dev), wet.to(dev), lfo.to(dev), None))
    with torch.no_grad():
        res = ol.tbptt_common_step(ref, None, dry, wet, lfo, 1024, 1024, {"l1": 1.0, "esr": 0.0, "dc": 0.0}, is_training=False)
    assert res["n_samples"] == 86410 and dd["wet_hat"].shape[-1] == 84992 == res["wet_hat"].shape[-1]
    assert float((dd["wet_hat"].cpu() - res["wet_hat"]).abs().max()) < 2e-5
    assert abs(float(loss) - float(res["loss"])) < 1e-6
    for k in ("l1", "esr", "dc"):
        assert abs(float(mod.logged[f"val/{k}"][-1]) - float(res["terms"][k])) < 1e-5 * max(1.0, abs(float(res["terms"][k])))


def test_streaming_effect_model_vs_reference_golden(golden_dir, dev):
    """(f) rank 4: four consecutive stereo buffers through the streaming effect model (LFO phase and LSTM state
    carried), against the real reference `EffectModel` / `do_forward_pass` (tests/golden/make_golden_streaming.py).
    LFO phase bookkeeping: exact; waveforms: 1e-5 (fp32, north_star)."""
    from mod_extraction_amd import streaming
    g = load(golden_dir, "streaming.npz")
    w = load(golden_dir, "lstm.npz")
    i = int(g["weights_index"])
    assert str(w[f"name_{i}"]) == str(g["weights_name"])
    em = streaming.EffectModel()
    em.model.load_state_dict({k[len(f"w_{i}_"):]: torch.from_numpy(w[k]) for k in w.files if k.startswith(f"w_{i}_")},
                             strict=True)
    em = em.to(dev)
    wrap = streaming.EffectModelWrapper(em)
    params = {k: torch.tensor(float(g[f"p_{k}"])) for k in ("lfo_rate", "lfo_depth", "lfo_stereo_phase_offset")}
    em.model.clear_hidden()
    for b, n in enumerate(g["sizes"]):
        x = torch.from_numpy(g[f"x{b}"]).to(dev)
        y = wrap.do_forward_pass(x, params)
        assert y.shape == (2, int(n))
        assert np.abs(y.cpu().numpy() - g[f"y{b}"]).max() < 1e-5, b
        assert float(em.prev_phase) == float(g[f"phase{b}"]), b            # carried LFO phase: bit-exact


class _FixedLFOExtractor(torch.nn.Module):
    """Stands in for a TRAINED frozen extractor (none exists in this image; an untrained CNN yields no valid LFO): runs
    the real CNN forward, then returns fixed per-row LFOs -- rows 0 and 2 valid sweeps, row 1 constant (no corner: discarded
    by the validity filter), row 3 a 40-cycle wiggle (too many corners: discarded)."""

    def __init__(self, net):
        super().__init__()
        self.net = net
        self.n_frames = net.n_frames

    def forward(self, x):
        hat, latent = self.net(x)
        t = torch.linspace(0.0, 1.0, hat.size(-1), device=hat.device)
        rows = torch.stack([0.5 + 0.5 * torch.cos(2 * math.pi * (1.5 * t + 0.1)), torch.full_like(t, 0.3),
                            0.5 + 0.5 * torch.cos(2 * math.pi * (2.5 * t + 0.6)), 0.5 + 0.5 * torch.cos(2 * math.pi * 40.0 * t)])
        return rows[: hat.size(0)].unsqueeze(1) + 0.0 * hat, latent


@pytest.mark.parametrize("yaml_flags", [False, True])
def test_tbptt_prefetched_prepare_is_bit_identical(dev, yaml_flags):
    """The effect-modelling trainer renders batch i+1 and runs the frozen extractor on it on a side stream while the LSTM
    trains on batch i (data_modules.set_ahead_fn / lightning.prepare_ahead).  Same weights after two batches, bit for bit,
    as with everything on the main stream.  yaml_flags: the shipped train_em_dry_wet.yml settings -- should_stretch and
    discard_invalid_lfos true -- where the prefetch must hand over the validity verdicts WITHOUT blocking the host
    (asynchronous copy behind an event) and the ragged batch (2 of 4 clips survive) is gathered when it is consumed."""
    from mod_extraction_amd import data_modules, lightning as al, models as am, optim, trainer

    def run(prefetch):
        torch.manual_seed(12); np.random.seed(12)
        cnn = am.Spectral2DCNN(in_ch=2, n_samples=22272, n_mels=64, out_channels=[64] * 6, temp_dilations=[1, 1, 2, 4, 8, 16],
                               pool_size=(2, 1))
        em = am.LSTMEffectModel()
        mod = al.TBPTTLFOEffectModeling(1024, 1024, em, lfo_model=_FixedLFOExtractor(cnn) if yaml_flags else cnn,
                                        discard_invalid_lfos=yaml_flags, should_stretch=True,
                                        loss_dict={"l1": 1.0, "esr": 0.0, "dc": 0.0}).to(dev).train()
        opt = optim.FlatAdamW([p for p in mod.parameters() if p.requires_grad], lr=1e-4, betas=(0.8, 0.99))
        dm = data_modules.RandomAudioChunkDryWetDataModule(batch_size=4, n_samples=22272, sr=44100, train_num_examples_per_epoch=8,
                                                           val_num_examples_per_epoch=4, overlap=prefetch)
        dm.setup(dev, rank=0, seed=77)
        if prefetch:
            t = trainer.Trainer(max_epochs=1, log_fn=None, limit_val_batches=1)
            hist = t.fit(mod, dm, opt)                                   # installs prepare_ahead, 2 train batches
            loss = hist[0]["train/loss"]
        else:
            for i in range(2):
                mod.training_step(dm.train_batch(), i, optimizer=opt, world_size=1)
            loss = float(torch.stack(mod.logged["train/loss"]).mean())
        if yaml_flags:
            assert mod.last_kept == 2                                    # rows 1 and 3 were discarded
        return opt.flat_param.clone(), opt.step_count, loss

    p0, n0, l0 = run(False)
    p1, n1, l1 = run(True)
    assert n0 == n1 == 2 * ((int((81 / 88) * 22272) - 1024) // 1024)
    assert torch.equal(p0, p1)
    assert abs(l0 - l1) < 1e-7


@pytest.mark.parametrize("weights,T", [({"l1": 0.5, "esr": 0.5}, 1024), ({"l1": 0.3, "mse": 0.7, "esr": 0.4, "dc": 2.0}, 1000),
                                       ({"mrstft": 1.0}, 4096), ({"mrstft": 0.7, "l1": 0.5, "esr": 0.2}, 4500)])
def test_lstm_bptt_any_loss_vs_autograd(dev, weights, T):
    """mx_lstm_bwd (upstream gradient d loss / d y from mx_effect_loss_grad / mx_mrstft_loss) against torch autograd
    through nn.LSTM with the oracle's loss modules (lightning.py:380-382 back-propagates any loss_dict).  LSTM gradients
    at the 1e-4 norm-wise gate of the fused-L1 test; d loss / d y itself at 1e-5 (MR-STFT: its fp32 gradient carries the
    division by bin magnitudes -- gate 2e-3 against the fp32 oracle as in test_gpu_mrstft.py, measured values printed)."""
    from mod_extraction_amd import effect_losses, models as am
    from oracle import losses as olosses
    torch.manual_seed(11)
    B = 3
    x = torch.rand(B, 1, 1024 + T) * 1.6 - 0.8
    lat = torch.rand(B, 1, 1024 + T)
    wet = (0.6 * x + 0.3 * torch.roll(x, 2, -1)).clamp(-1, 1)
    sd = om.LSTMEffectModel(1, 1, 64, 1).state_dict()
    ref = om.LSTMEffectModel(1, 1, 64, 1); ref.load_state_dict(sd)
    mine = am.LSTMEffectModel(1, 1, 64, 1); mine.load_state_dict(sd); mine = mine.to(dev)
    ref.clear_hidden(); ref(x[..., :1024], lat[..., :1024]); ref.detach_hidden()
    y_r = ref(x[..., 1024:], lat[..., 1024:])
    y_r.retain_grad()
    loss_r = sum(w * olosses.get_loss_func_by_name(k)(y_r, wet[..., 1024:]) for k, w in weights.items())
    loss_r.backward()
    xd, ld, wd = x.to(dev), lat.to(dev), wet.to(dev)
    mine.clear_hidden(); mine.run_chunk(xd[..., :1024], ld[..., :1024]); mine.detach_hidden()
    stash = torch.empty((B, T, 384), device=dev)
    y_m, h0, c0 = mine.run_chunk(xd[..., 1024:], ld[..., 1024:], stash)
    dy = effect_losses.effect_loss_grad(y_m, wd[..., 1024:].contiguous(), weights)
    e_dy = float((dy.cpu() - y_r.grad[:, 0]).abs().max() / y_r.grad.abs().max())
    print(f"[measured] d loss / d y ({'+'.join(weights)}, T={T}): rel err {e_dy:.2e}")
    assert e_dy < (2e-3 if "mrstft" in weights else 1e-5), e_dy
    grad = torch.empty(am.LSTM_NPARAM, device=dev)
    mine.bptt_chunk(xd[..., 1024:], ld[..., 1024:], y_m, dy, stash, h0, c0, grad)
    off = 0
    for n, p in ref.named_parameters():
        k = p.numel()
        a, r = grad[off:off + k].cpu(), p.grad.reshape(-1)
        e = float((a - r).abs().max() / r.abs().max())
        assert e < (2e-3 if "mrstft" in weights else 1e-4), (n, e)
        off += k
    assert off == am.LSTM_NPARAM


def test_lstm_bptt_general_path_equals_the_fused_l1_path(dev):
    """With only nn.L1Loss weighted, mx_effect_loss_grad + mx_lstm_bwd give the bits of the fused mx_lstm_bwd_l1."""
    from mod_extraction_amd import effect_losses, models as am
    torch.manual_seed(12)
    B, T = 4, 777
    x, lat = torch.rand(B, 1, T, device=dev) * 1.6 - 0.8, torch.rand(B, 1, T, device=dev)
    wet = (0.5 * x).contiguous()
    em = am.LSTMEffectModel().to(dev)
    em.clear_hidden()
    stash = torch.empty((B, T, 384), device=dev)
    y, h0, c0 = em.run_chunk(x, lat, stash)
    g_fused, g_gen = torch.empty(am.LSTM_NPARAM, device=dev), torch.empty(am.LSTM_NPARAM, device=dev)
    em.bptt_l1_chunk(x, lat, y, wet, stash, h0, c0, 0.25 / (B * T), g_fused)
    em.bptt_chunk(x, lat, y, effect_losses.effect_loss_grad(y, wet, {"l1": 0.25, "esr": 0.0}), stash, h0, c0, g_gen)
    assert torch.equal(g_fused, g_gen)


@pytest.mark.parametrize("ld,W,S,n", [({"l1": 0.5, "esr": 0.5, "dc": 0.0}, 1024, 1024, 6000),
                                      ({"mrstft": 1.0, "l1": 0.5}, 1024, 4096, 14000)])
def test_tbptt_training_with_other_losses_vs_oracle(dev, ld, W, S, n):
    """TBPTTLFOEffectModeling with esr / dc / mrstft weighted (the reference trains on whatever loss_dict holds,
    lightning.py:380-382; BASELINE config 4 is worded '+ MR-STFT loss'): optimizer steps, wet_hat, logged terms and the
    weights after the steps against the oracle running torch autograd + torch.optim.AdamW."""
    from mod_extraction_amd import lightning as al, models as am, optim
    from oracle import modulations as omod
    torch.manual_seed(W + S)
    B = 3
    dry = torch.rand(B, 1, n) * 1.6 - 0.8
    wet = (0.7 * dry + 0.2 * torch.roll(dry, 5, -1)).clamp(-1, 1)
    lfo = torch.stack([omod.make_mod_signal(64, 64 / (n / 44100.0), f, p, "cos") for f, p in ((6.0, 0.2), (9.0, 1.0), (7.5, 3.0))])
    ref = om.LSTMEffectModel()
    init = {k: v.clone() for k, v in ref.state_dict().items()}
    em = am.LSTMEffectModel(); em.load_state_dict(init)
    mod = al.TBPTTLFOEffectModeling(W, S, em, lfo_model=None, model_smooth_n_frames=0, should_stretch=False,
                                    discard_invalid_lfos=False, loss_dict=ld).to(dev).train()
    opt = optim.FlatAdamW(mod.parameters(), lr=1e-3, betas=(0.8, 0.99))
    loss, dd, _ = mod.common_step((dry.to(dev), wet.to(dev), lfo.to(dev), None), is_training=True, optimizer=opt, world_size=1)
    ropt = torch.optim.AdamW(ref.parameters(), lr=1e-3, betas=(0.8, 0.99))
    res = ol.tbptt_common_step(ref, ropt, dry, wet, lfo, W, S, ld, is_training=True, model_smooth_n_frames=0,
                               should_stretch=False, discard_invalid_lfos=False)
    assert opt.step_count == res["steps"] == (n - W) // S
    assert float((dd["wet_hat"].cpu() - res["wet_hat"]).abs().max()) < 1e-4
    assert abs(float(loss) - float(res["loss"])) < 1e-5 * max(1.0, abs(float(res["loss"])))
    for k in ld:
        assert abs(float(mod.logged[f"train/{k}"][-1]) - float(res["terms"][k])) < 1e-5 * max(1.0, abs(float(res["terms"][k]))), k
    for k, v in em.state_dict().items():
        d = (v.cpu() - ref.state_dict()[k]).abs()
        moved = (ref.state_dict()[k] - init[k]).abs()
        assert float(d.median()) < 0.02 * max(float(moved.median()), 1e-9), k


def test_reduce_rows_adamw_step_is_the_two_launches_in_one(dev):
    """mx_reduce_rows_adamw_step (the TBPTT loop's 83 optimizer steps per batch, one process, frozen extractor) against
    mx_reduce_rows followed by mx_adamw_step on the same rows: flat gradient, parameters and both moments bit-identical
    over three steps (same fp64 column sums in the same order, same update expression)."""
    from mod_extraction_amd import _hip, optim
    torch.manual_seed(3)
    n, rows = 17473, 128
    pa = [torch.nn.Parameter(torch.randn(n, device=dev) * 0.1)]
    pb = [torch.nn.Parameter(pa[0].detach().clone())]
    oa, ob = optim.FlatAdamW(pa, lr=1e-3, betas=(0.8, 0.99)), optim.FlatAdamW(pb, lr=1e-3, betas=(0.8, 0.99))
    for step in range(3):
        part = torch.randn(rows, n, device=dev) * (10.0 ** -step)
        _hip.call("mx_reduce_rows", _hip.ptr(part), rows, n, 0, _hip.ptr(oa.flat_grad), _hip.stream())
        oa.step(grad_scale=0.5)
        ob.step_from_rows(part, grad_scale=0.5)
        assert torch.equal(oa.flat_grad, ob.flat_grad)
        assert torch.equal(oa.flat_param, ob.flat_param)
        assert torch.equal(oa.exp_avg, ob.exp_avg) and torch.equal(oa.exp_avg_sq, ob.exp_avg_sq)
    assert oa.step_count == ob.step_count == 3
