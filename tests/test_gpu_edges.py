"""GPU parity at the edges of the kernels' internal tilings (ragged and minimum sizes), against the CPU oracle:
LSTM-64 chunk lengths around the 32-step stash slabs and the 256-step output blocks, batch 1; MR-STFT clips whose
frame counts are not multiples of the 16 frames a workgroup (or the 1 / 2 frames a wavefront) takes, and the shortest
clip reflect padding allows; flanger clips that end inside a 256-sample chunk.  Tolerances as in the kernels' own test
files (audio 1e-5 absolute, LSTM gradients 1e-4 of each tensor's max, MR-STFT value 1e-5 relative / gradient against the
fp64 oracle, flanger bit-exact)."""
import numpy as np
import pytest
import torch

from oracle import fx as ofx, losses as olosses, models as om, modulations as omod

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,T", [(1, 1), (2, 2), (1, 31), (2, 33), (1, 64), (2, 255), (1, 257), (3, 300), (1, 513),
                                 # round 6: the forward's unrolled groups of 16 and the backward's unrolled 32-step slabs, alone and
                                 # next to a partial group / slab, a one-slab chunk, a second 256-step block of one group
                                 (1, 16), (2, 32), (1, 48), (2, 272), (1, 1056)])
def test_lstm_chunk_lengths_around_the_slab_and_block_sizes(dev, B, T):
    from mod_extraction_amd import models as am
    torch.manual_seed(100 * B + T)
    ref = om.LSTMEffectModel(1, 1, 64, 1)
    mine = am.LSTMEffectModel(1, 1, 64, 1)
    mine.load_state_dict(ref.state_dict())
    mine = mine.to(dev)
    W = 40                                               # warm-up chunk: a non-zero state enters the chunk under test
    x = torch.rand(B, 1, W + T) * 1.6 - 0.8
    lat = torch.rand(B, 1, W + T)
    wet = (0.6 * x + 0.3 * torch.roll(x, 2, -1)).clamp(-1, 1)
    ref.clear_hidden(); ref(x[..., :W], lat[..., :W]); ref.detach_hidden()
    y_r = ref(x[..., W:], lat[..., W:])
    torch.nn.functional.l1_loss(y_r, wet[..., W:]).backward()
    xd, ld, wd = x.to(dev), lat.to(dev), wet.to(dev)
    mine.clear_hidden(); mine.run_chunk(xd[..., :W], ld[..., :W]); mine.detach_hidden()
    stash = torch.empty((B, T, 384), device=dev)
    y_m, h0, c0 = mine.run_chunk(xd[..., W:], ld[..., W:], stash)
    assert float((y_m.cpu() - y_r.detach()).abs().max()) < 1e-5
    assert float((mine.hidden[0].cpu() - ref.hidden[0].detach()).abs().max()) < 1e-5
    assert float((mine.hidden[1].cpu() - ref.hidden[1].detach()).abs().max()) < 2e-5
    # the inference variant of the forward kernel (no stash) gives the same output
    mine.clear_hidden(); mine.run_chunk(xd[..., :W], ld[..., :W]); mine.detach_hidden()
    y_i = mine.run_chunk(xd[..., W:], ld[..., W:])[0]
    assert float((y_i - y_m).abs().max()) < 1e-6
    grad = torch.empty(am.LSTM_NPARAM, device=dev)
    mine.bptt_l1_chunk(xd[..., W:], ld[..., W:], y_m, wd[..., W:], stash, h0, c0, 1.0 / (B * T), grad)
    off = 0
    for n, p in ref.named_parameters():
        k = p.numel()
        a, r = grad[off:off + k].cpu(), p.grad.reshape(-1)
        e = float((a - r).abs().max() / r.abs().max().clamp_min(1e-12))
        assert e < 1e-4, (n, e)
        off += k


@pytest.mark.parametrize("B,T", [(1, 1025), (1, 1100), (2, 1999), (1, 4097), (3, 3851)])
def test_mrstft_ragged_frame_counts_and_shortest_clip(dev, B, T):
    """T = 1025 is the shortest clip the 2048-point resolution's reflect padding accepts (pad 1024 < T); the others leave
    frame counts that are not multiples of 16 (frames per workgroup) or 2 (frames per wavefront at n_fft 512)."""
    from mod_extraction_amd import losses as alosses
    torch.manual_seed(B + T)
    t = torch.arange(T) / 44100.0
    y = (0.5 * torch.sin(2 * np.pi * 330.0 * t) + 0.2 * torch.rand(B, 1, T) - 0.1).clamp(-1, 1)
    x = (0.8 * y + 0.1 * torch.roll(y, 7, -1) + 0.05 * torch.randn(B, 1, T)).clamp(-1, 1)

    class MR64(olosses.MultiResolutionSTFTLoss):
        def _mag(self, v, n_fft, hop, win):
            s = torch.stft(v.reshape(-1, v.size(-1)), n_fft, hop, win, torch.hann_window(win, dtype=torch.float64),
                           return_complex=True)
            return torch.sqrt(torch.clamp(s.real ** 2 + s.imag ** 2, min=self.eps))
    x64 = x.double().requires_grad_(True)
    loss64 = MR64()(x64, y.double())
    loss64.backward()
    x32 = x.clone().requires_grad_(True)
    loss32 = olosses.get_loss_func_by_name("mrstft")(x32, y)
    loss32.backward()
    xd = x.to(dev).requires_grad_(True)
    loss_m = alosses.get_loss_func_by_name("mrstft")(xd, y.to(dev))
    loss_m.backward()
    assert abs(float(loss_m) - float(loss64)) < 1e-5 * abs(float(loss64)), (float(loss_m), float(loss64))
    scale = x64.grad.abs().max()
    e_mine = float((xd.grad.cpu().double() - x64.grad).abs().max() / scale)
    e_oracle32 = float((x32.grad.double() - x64.grad).abs().max() / scale)
    assert e_mine < 2e-3 and e_mine < max(2.0 * e_oracle32, 1e-4), (e_mine, e_oracle32)


@pytest.mark.parametrize("B,N", [(1, 1), (1, 255), (2, 257), (3, 1000), (1, 4099)])
def test_flanger_clip_lengths_inside_a_chunk_bit_exact(dev, B, N):
    from mod_extraction_amd import fx as afx
    torch.manual_seed(7 * B + N)
    x = torch.rand(B, N) * 2 - 1
    mod = torch.stack([omod.make_mod_signal(N, 44100, 0.7 + 1.3 * i, 0.4 * i, "cos") for i in range(B)])
    for mm, ml in ((1.0, 10.0), (30.0, 10.0)):
        Mm, Ml = ofx.delay_samples(mm, 44100), ofx.delay_samples(ml, 44100)
        p = dict(feedback=torch.rand(B) * 0.7, min_delay_width=torch.rand(B), width=torch.rand(B),
                 depth=torch.rand(B), mix=torch.rand(B))
        po = ofx.derive_params(B, Mm, Ml, **p)
        y_ref = ofx.flanger_np(x.numpy(), mod.numpy(), po, Mm + Ml)
        y_ref = y_ref[0] if isinstance(y_ref, tuple) else y_ref
        consts = afx.derive_clip_constants(B, dev, Mm, Ml, **{k: v.to(dev) for k, v in p.items()})
        md = torch.full((B,), Mm + Ml, dtype=torch.int32, device=dev)
        y = afx.flanger_forward(x.to(dev), mod.to(dev), consts, md, Mm + Ml)
        assert np.array_equal(y.cpu().numpy(), y_ref), (N, mm, np.abs(y.cpu().numpy() - y_ref).max())


@pytest.mark.parametrize("N", [1, 3, 63, 65, 255, 257, 1000])
def test_phaser_clip_lengths_inside_a_block(dev, N):
    """lead + N ending inside a 4-sample cut-off update, a 64-sample sub-block and a 256-sample block of the producer /
    consumer kernel; both kernels (state-space and JUCE operation order) against oracle_ref.c:orc_phaser at 1e-5."""
    from mod_extraction_amd import fx as afx
    torch.manual_seed(N)
    lead = torch.tensor([0, 1, 62, 259], dtype=torch.int32)
    B = lead.numel()
    src = torch.rand(B, N + 259) * 1.6 - 0.8
    p = {"rate_hz": torch.tensor([0.5, 3.0, 1.3, 2.2]), "depth": torch.tensor([1.0, 0.2, 0.6, 0.9]),
         "centre_frequency_hz": torch.tensor([70.0, 18000.0, 440.0, 1300.0]),
         "feedback": torch.tensor([0.0, 0.7, 0.25, 0.5]), "mix": torch.tensor([1.0, 0.2, 0.5, 0.8])}
    pd = {k: v.to(dev) for k, v in p.items()}
    y = torch.empty(B, N, device=dev)
    y_exact = torch.empty(B, N, device=dev)
    dry = torch.empty(B, N, device=dev)
    afx.phaser_forward(src.to(dev), pd, lead.to(dev), 44100.0, N, out=y, dry_out=dry)
    afx.phaser_forward(src.to(dev), pd, lead.to(dev), 44100.0, N, out=y_exact, exact_order=True)
    for b in range(B):
        L = int(lead[b])
        ref = ofx.phaser_np(src[b:b + 1, :L + N].numpy(), [float(p["rate_hz"][b])], [float(p["depth"][b])],
                            [float(p["centre_frequency_hz"][b])], [float(p["feedback"][b])], [float(p["mix"][b])], 44100.0)
        assert np.array_equal(dry[b].cpu().numpy(), src[b, L:L + N].numpy())
        for name, out in (("state-space", y), ("juce-order", y_exact)):
            err = np.abs(out[b].cpu().numpy() - ref[0, L:]).max()
            assert err < 1e-5, (name, b, N, err)


@pytest.mark.parametrize("W,S,n", [(512, 256, 6000), (100, 37, 3000), (1024, 1024, 5000)])
def test_tbptt_training_with_other_warmup_and_step_lengths(dev, W, S, n):
    """TBPTT training batch with warm-up != step length, step lengths off the kernels' slab sizes and a tail that does
    not fill a step, against the oracle running torch.optim.AdamW on the CPU: kept clips, optimizer steps, wet_hat, losses
    and the weights after the steps (bulk statistic: Adam is ill-conditioned where a gradient is ~eps)."""
    from mod_extraction_amd import lightning as al, models as am, optim
    from oracle import lightning as ol
    torch.manual_seed(W + S)
    B = 3
    dry = torch.rand(B, 1, n) * 1.6 - 0.8
    wet = (0.7 * dry + 0.2 * torch.roll(dry, 5, -1)).clamp(-1, 1)
    frames = 64
    lfo = torch.stack([omod.make_mod_signal(frames, frames / (n / 44100.0), f, p, "cos")
                       for f, p in ((6.0, 0.2), (9.0, 1.0), (7.5, 3.0))])
    ref = om.LSTMEffectModel()
    init = {k: v.clone() for k, v in ref.state_dict().items()}
    em = am.LSTMEffectModel(); em.load_state_dict(init)
    ld = {"l1": 1.0, "esr": 0.0, "dc": 0.0}
    mod = al.TBPTTLFOEffectModeling(W, S, em, lfo_model=None, model_smooth_n_frames=0, should_stretch=False,
                                    discard_invalid_lfos=False, loss_dict=ld).to(dev).train()
    opt = optim.FlatAdamW(mod.parameters(), lr=1e-3, betas=(0.8, 0.99))
    loss, dd, _ = mod.common_step((dry.to(dev), wet.to(dev), lfo.to(dev), None), is_training=True, optimizer=opt, world_size=1)
    ropt = torch.optim.AdamW(ref.parameters(), lr=1e-3, betas=(0.8, 0.99))
    res = ol.tbptt_common_step(ref, ropt, dry, wet, lfo, W, S, ld, is_training=True, model_smooth_n_frames=0,
                               should_stretch=False, discard_invalid_lfos=False)
    assert opt.step_count == res["steps"] == (n - W) // S
    assert dd["wet_hat"].shape == res["wet_hat"].shape
    assert float((dd["wet_hat"].cpu() - res["wet_hat"]).abs().max()) < 1e-4
    assert abs(float(loss) - float(res["loss"])) < 1e-5
    for k, v in em.state_dict().items():
        want = ref.state_dict()[k]
        d = (v.cpu() - want).abs()
        moved = (want - init[k]).abs()
        assert float(d.median()) < 0.02 * max(float(moved.median()), 1e-9), k


@pytest.mark.parametrize("over,n_samples", [
    (dict(hop_len=128, n_mels=64), 30000),
    (dict(hop_len=512, n_mels=96, sr=22050, out_channels=[64] * 5, temp_dilations=[1, 2, 4, 8, 16]), 88200),
    (dict(hop_len=300, n_mels=128, sr=48000), 48000),
    (dict(n_fft=512, hop_len=128, n_mels=64), 40000),                     # the other transform sizes of the constructor
    (dict(n_fft=2048, hop_len=512, n_mels=128), 88200),
    (dict(n_fft=2048, hop_len=256, n_mels=256), 30001),
    (dict(n_fft=512, hop_len=100, n_mels=64, sr=16000), 16000),
    # 59 KB static + 49 KB dynamic LDS: above 64 KB, needs the per-device dynamic-LDS attribute
    (dict(n_fft=2048, hop_len=512, n_mels=512, out_channels=[64] * 7, temp_dilations=[1, 1, 2, 4, 8, 16, 1]), 88200),
])
def test_logmel_other_hops_rates_and_band_counts(dev, over, n_samples):
    """mel front end away from the shipped (hop 256, 256 bands, 44.1 kHz): frame count, filter bank and framing follow
    the arguments, values against the oracle's torchaudio restatement at the log-mel tolerance of tests/test_gpu_cnn.py."""
    from mod_extraction_amd import models as amodels
    cfg = dict(in_ch=2, n_samples=n_samples, sr=44100, n_fft=1024, hop_len=256, n_mels=256, kernel_size=(5, 13),
               out_channels=[64] * 6, temp_dilations=[1, 1, 2, 4, 8, 16], pool_size=(2, 1), latent_dim=1, use_ln=True)
    cfg.update(over)
    torch.manual_seed(0)
    ref = om.Spectral2DCNN(**cfg)
    mine = amodels.Spectral2DCNN(**cfg)
    mine.load_state_dict(ref.state_dict(), strict=True)
    mine = mine.to(dev)
    g = torch.Generator().manual_seed(3)
    x = torch.rand(2, 2, n_samples, generator=g) * 1.6 - 0.8
    masks = (1, 5, 3, 17)
    with torch.no_grad():
        want = ref.log_mel(x, masks)
        got = mine.log_mel(x.to(dev), masks).cpu()
    n_frames = n_samples // cfg["hop_len"] + 1
    assert want.shape[-1] == n_frames and got.shape[:3] == want.shape[:3]
    assert torch.all(got[..., n_frames:] == 0)
    got = got[..., :n_frames]
    floor = np.log(1e-7)
    live = want > floor + 1.0
    err = (got - want).abs()
    # narrow filters (512 bands over 1025 bins: single-bin bands) pass one bin's power through unaveraged, and a bin that
    # holds 1e-5 of the frame's power is only known to ~1e-2 in fp32 on EITHER side: there an fp64 evaluation of the oracle
    # arbitrates (as for the MR-STFT gradients) -- the kernel may be as far from it as 3x the fp32 oracle is, or inside the gate
    import copy
    want64 = copy.deepcopy(ref).double().log_mel(x.double(), masks)
    e_mine = (got.double() - want64)[live].abs().max()
    e_orc32 = (want.double() - want64)[live].abs().max()
    gate = 2e-5 + 1e-5 * float(want[live].abs().max())
    assert float(e_mine) <= max(gate, 3.0 * float(e_orc32)), (float(e_mine), float(e_orc32))
    if cfg["n_mels"] <= 256:
        assert float(err[live].max()) <= gate, float(err[live].max())
    assert float(err[~live].max()) < 0.05 if (~live).any() else True


def test_logmel_band_count_beyond_the_lds_budget_is_unsupported_not_a_failed_launch(dev):
    """mx_logmel_fwd sizes its LDS from n_mels: a band count whose mel tile + filter coefficients do not fit the CU's
    160 KB beside the FFT buffers must come back as MX_ERR_UNSUPPORTED before any launch (it used to fail AT launch)."""
    from mod_extraction_amd import _hip
    n_fft, n_mels, N, hop = 2048, 2048, 30000, 512
    n_frames = N // hop + 1
    z = lambda *s, dt=torch.float32: torch.zeros(*s, device=dev, dtype=dt)
    x, win, tw = z(2, N), z(n_fft), z(n_fft, 2)
    fb, lo, hi = z(n_fft // 2 + 1, n_mels), z(n_mels, dt=torch.int32), z(n_mels, dt=torch.int32)
    out = z(2, n_mels, n_frames)
    args = lambda m: (_hip.ptr(x), 2, N, _hip.ptr(win), _hip.ptr(tw), _hip.ptr(fb), _hip.ptr(lo), _hip.ptr(hi), n_fft, hop, m,
                      n_frames, n_frames, 1e-7, 0, 0, 0, 0, _hip.ptr(out), _hip.stream())
    assert _hip.load().mx_logmel_fwd(*args(n_mels)) == -2                 # MX_ERR_UNSUPPORTED
    assert _hip.load().mx_logmel_fwd(*args(1024)) == 0                    # 59 + 90 KB: fits, with the attribute
    torch.cuda.synchronize()


@pytest.mark.parametrize("kw", [dict(lr=3e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.1),
                                dict(lr=1e-2, betas=(0.5, 0.9), eps=1e-8, weight_decay=0.0)])
def test_adamw_kernel_other_hyperparameters(dev, kw):
    """mx_adamw_step against torch.optim.AdamW away from configs/opt/adam_w.yml; 33 parameters so that the flat buffer
    ends inside a vector / a wavefront, 40 steps so that the bias corrections run through their steep part."""
    from mod_extraction_amd import optim
    torch.manual_seed(1)
    ps = [torch.nn.Parameter(torch.randn(7, 3)), torch.nn.Parameter(torch.randn(11)), torch.nn.Parameter(torch.randn(1))]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    mine = [torch.nn.Parameter(p.detach().clone().to(dev)) for p in ps]
    o_ref = torch.optim.AdamW(ref, **kw)
    o_mine = optim.FlatAdamW(mine, **kw)
    for step in range(40):
        grads = [torch.randn_like(p) * (0.5 ** (step % 9)) for p in ps]
        for p, g in zip(ref, grads):
            p.grad = g.clone()
        o_mine.zero_grad()
        for p, g in zip(mine, grads):
            p.grad.copy_(g.to(dev))
        o_ref.step()
        o_mine.step()
        for p, q in zip(mine, ref):
            assert float((p.detach().cpu() - q.detach()).abs().max()) < 2e-6 * max(1.0, float(q.detach().abs().max())), step


@pytest.mark.parametrize("use_dry,smooth,ld", [
    (False, 4, {"l1": 1.0, "fdl1": 0.0, "sdl1": 2.0, "mse": 0.5}),
    (True, 8, {"l1": 0.0, "mse": 1.0}),
    (True, 0, {"l1": 1.0, "fdl1": 5.0, "sdl1": 10.0, "mse": 0.25}),
])
def test_lfo_extraction_train_step_other_options(dev, use_dry, smooth, ld):
    """LFOExtraction training away from train_lfo_*.yml: wet-only input, output smoothing inside the differentiated path,
    other loss mixes; loss terms and every parameter gradient against the oracle's autograd."""
    from mod_extraction_amd import lightning, models, optim
    from oracle import lightning as ol
    n, sr, B = 22272, 44100, 4
    cfg = dict(in_ch=2 if use_dry else 1, n_samples=n, sr=sr, n_fft=1024, hop_len=256, n_mels=64, kernel_size=(5, 13),
               out_channels=[64] * 6, temp_dilations=[1, 1, 2, 4, 8, 16], pool_size=(2, 1), latent_dim=1, use_ln=True)
    torch.manual_seed(smooth)
    ref = om.Spectral2DCNN(**cfg).eval()
    mine = models.Spectral2DCNN(**cfg)
    mine.load_state_dict(ref.state_dict())
    module = lightning.LFOExtraction(mine, sr=sr, use_dry=use_dry, model_smooth_n_frames=smooth, should_stretch=False,
                                     loss_dict=ld).to(dev).eval()
    opt = optim.FlatAdamW(module.parameters(), lr=1e-4, betas=(0.8, 0.99))
    dry = torch.rand(B, 1, n) * 1.6 - 0.8
    wet = (0.6 * dry + 0.3 * torch.roll(dry, 9, -1)).clamp(-1, 1)
    mod = torch.stack([omod.make_mod_signal(882, 441.0, 0.8 + 0.7 * i, 0.5 * i, "cos") for i in range(B)])
    loss_r, terms_r, _ = ol.lfo_common_step(ref, dry, wet, mod, ld, use_dry=use_dry, model_smooth_n_frames=smooth)
    loss_r.backward()
    opt.zero_grad()
    loss = module.training_step((dry.to(dev), wet.to(dev), mod.to(dev), None), 0)
    loss.backward()
    assert abs(float(loss.detach()) - float(loss_r.detach())) < 1e-5 * max(1.0, abs(float(loss_r.detach())))
    for k in ld:
        assert abs(float(module.logged[f"train/{k}"][-1]) - float(terms_r[k].detach())) < 2e-6, k
    # gradients: the loss kernel's d/d y_hat is compared below and the CNN's backward in tests/test_gpu_cnn.py with the
    # device's pooling / PReLU decisions routed into the oracle (un-routed, ties decided by fp32 rounding differ by ~1e-3);
    # here: every parameter received a finite, non-trivial gradient through the smoothing and the loss mix
    for name, p in mine.named_parameters():
        assert torch.isfinite(p.grad).all() and float(p.grad.abs().max()) > 0, name


@pytest.mark.parametrize("w", [{"l1": 1.0, "fdl1": 0.0, "sdl1": 2.0, "mse": 0.5}, {"l1": 0.0, "mse": 1.0},
                               {"l1": 1.0, "fdl1": 5.0, "sdl1": 10.0, "mse": 0.25}, {"mse": 2.0, "sdl1": 1.0}])
@pytest.mark.parametrize("n", [2, 3, 4, 5, 64, 65, 345])
def test_lfo_loss_kernel_other_mixes_and_lengths(dev, w, n):
    """mx_lfo_loss (values and d / d y_hat in one launch) for loss mixes other than train_lfo_*.yml and rows as short as
    the central differences allow: losses.py's central_diff asserts more than 2 points, i.e. fdl1 needs n >= 3 and sdl1
    n >= 5 -- below that the reference raises, and so must the product."""
    from mod_extraction_amd import losses as alosses
    torch.manual_seed(n)
    y_hat = torch.rand(5, n, requires_grad=True)
    y = torch.rand(5, n)
    try:
        terms_r = {k: olosses.get_loss_func_by_name(k)(y_hat, y) for k in w}
    except AssertionError:
        with pytest.raises(Exception):
            alosses.lfo_loss(y_hat.detach().to(dev), y.to(dev), w)
        return
    tot_r = sum(w[k] * terms_r[k] for k in w if w[k] > 0)
    finite = bool(torch.isfinite(tot_r))
    if finite:
        tot_r.backward()
    yh = y_hat.detach().to(dev).requires_grad_(True)
    tot_m, terms_m = alosses.lfo_loss(yh, y.to(dev), w)
    for k in w:
        a, b = float(terms_m[k]), float(terms_r[k].detach())
        assert (np.isnan(a) and np.isnan(b)) or abs(a - b) < 2e-6, (k, a, b)
    if finite:
        tot_m.backward()
        assert abs(float(tot_m.detach()) - float(tot_r.detach())) < 1e-6 * max(1.0, abs(float(tot_r.detach())))
        assert float((yh.grad.cpu() - y_hat.grad).abs().max()) < 1e-5 * float(y_hat.grad.abs().max())


def test_log_mel_l1_metric(dev):
    """losses.py:105-130 / get_loss_func_by_name("log_mel_l1"): value against the oracle; forward-only by contract."""
    from mod_extraction_amd import losses as alosses
    torch.manual_seed(11)
    y = torch.rand(3, 1, 30000) * 1.6 - 0.8
    x = (0.8 * y + 0.1 * torch.roll(y, 4, -1)).clamp(-1, 1)
    want = float(olosses.get_loss_func_by_name("log_mel_l1")(x, y))
    fn = alosses.get_loss_func_by_name("log_mel_l1")
    got = float(fn(x.to(dev), y.to(dev)))
    assert abs(got - want) < 1e-5 * max(1.0, abs(want)), (got, want)
    assert float(fn(y.to(dev), y.to(dev))) == 0.0
    with pytest.raises(NotImplementedError):
        fn(x.to(dev).requires_grad_(True), y.to(dev))


def test_random_chunk_and_mod_sig_data_module_with_the_random_lfo_baseline(dev):
    """configs/eval_lfo_rand.yml's object graph: RandomAudioChunkAndModSigDataModule (data_modules.py:331-371) hands out
    (None, unprocessed chunk, random LFO label, fx_params); LFOExtraction with models.RandomLFO and use_dry false scores a
    perturbed-ground-truth guess against the label."""
    from mod_extraction_amd import data_modules, lightning, models
    torch.manual_seed(5); np.random.seed(5)
    dm = data_modules.RandomAudioChunkAndModSigDataModule(
        batch_size=5, n_samples=88200, sr=44100, val_num_examples_per_epoch=10,
        fx_config={"mod_sig": {"rate_hz": {"min": 0.5, "max": 3.0}, "phase": {"min": 0.0, "max": 6.28318530718},
                               "shapes": ["cos", "tri", "rect_cos", "inv_rect_cos", "saw", "rsaw"], "exp": 1.0}})
    dm.setup(dev, rank=0, seed=5)
    dry, wet, mod, params = dm.val_batch()
    assert dry is None and wet.shape == (5, 1, 88200) and mod.shape == (5, 882)
    assert float(wet.abs().max()) > 0.1                                   # an unprocessed chunk, not silence
    for i in range(5):                                                    # the label is the LFO of the drawn parameters
        ref = omod.make_mod_signal(882, 441.0, float(params["rate_hz"][i]), float(params["phase"][i]), params["shape"][i])
        assert float((mod[i].cpu() - ref).abs().max()) < 2e-5
    base = models.RandomLFO(345, 172.5, use_shape_gt=True, use_phase_gt=True, use_freq_gt=True, freq_min=0.5, freq_max=3.0,
                            phase_error=0.0, freq_error=0.0)
    module = lightning.LFOExtraction(base, sr=44100, use_dry=False, model_smooth_n_frames=0, should_stretch=False,
                                     loss_dict={"l1": 1.0, "mse": 0.0}).to(dev).eval()
    loss, data, _ = module.common_step((dry, wet, mod, params), is_training=False)
    assert data["mod_sig_hat"].shape == (5, 345)
    # ground-truth shape / phase / rate and no error: the guess is the label up to the two sampling grids (345 points at
    # 172.5 Hz against 882 points at 441 Hz resampled), which differ around the jumps of the saw shapes
    assert float(loss) < 0.02
    noisy = models.RandomLFO(345, 172.5, use_shape_gt=True, use_phase_gt=True, use_freq_gt=True, phase_error=0.5,
                             freq_error=0.25)
    module2 = lightning.LFOExtraction(noisy, sr=44100, use_dry=False, model_smooth_n_frames=0,
                                      loss_dict={"l1": 1.0, "mse": 0.0}).to(dev).eval()
    assert float(module2.common_step((dry, wet, mod, params), is_training=False)[0]) > float(loss)


def test_lfo_extraction_sub_batch_size_path(dev):
    """lightning.py:160-185: with sub_batch_size the step runs the extractor on slices of the batch and averages their
    losses.  Equal slices => the same loss and the same parameter gradients as the whole batch (same device decisions
    on both paths), and the loss of the oracle's whole-batch step."""
    from mod_extraction_amd import lightning, models, optim
    from oracle import lightning as ol
    n, sr, B = 22272, 44100, 4
    cfg = dict(in_ch=2, n_samples=n, sr=sr, n_fft=1024, hop_len=256, n_mels=64, kernel_size=(5, 13),
               out_channels=[64] * 6, temp_dilations=[1, 1, 2, 4, 8, 16], pool_size=(2, 1), latent_dim=1, use_ln=True)
    ld = {"l1": 1.0, "fdl1": 5.0, "sdl1": 10.0, "mse": 0.0}
    torch.manual_seed(21)
    ref = om.Spectral2DCNN(**cfg).eval()
    dry = torch.rand(B, 1, n) * 1.6 - 0.8
    wet = (0.6 * dry + 0.3 * torch.roll(dry, 9, -1)).clamp(-1, 1)
    mod = torch.stack([omod.make_mod_signal(882, 441.0, 0.8 + 0.7 * i, 0.5 * i, "cos") for i in range(B)])
    params = {"rate_hz": torch.rand(B), "shape": ["cos"] * B}
    grads, losses = [], []
    for sub in (None, 2, 1):
        mine = models.Spectral2DCNN(**cfg)
        mine.load_state_dict(ref.state_dict())
        module = lightning.LFOExtraction(mine, sr=sr, use_dry=True, model_smooth_n_frames=0, should_stretch=False,
                                         sub_batch_size=sub, loss_dict=ld).to(dev).eval()
        opt = optim.FlatAdamW(module.parameters(), lr=1e-4, betas=(0.8, 0.99))
        opt.zero_grad()
        loss = module.training_step((dry.to(dev), wet.to(dev), mod.to(dev), {"rate_hz": params["rate_hz"].to(dev),
                                                                            "shape": params["shape"]}), 0)
        loss.backward()
        grads.append(opt.flat_grad.clone())
        losses.append(float(loss.detach()))
        if sub is not None:
            assert len(module.logged["train/loss"]) == B // sub           # one log entry per slice
    loss_r, _, _ = ol.lfo_common_step(ref, dry, wet, mod, ld)
    for l in losses:
        assert abs(l - float(loss_r.detach())) < 1e-5 * max(1.0, abs(float(loss_r.detach())))
    scale = float(grads[0].abs().max())
    for g in grads[1:]:
        assert float((g - grads[0]).abs().max()) < 2e-5 * scale


@pytest.mark.parametrize("kinds", [("flanger",), ("chorus",), ("phaser",), ("dry",)])
def test_batcher_with_a_single_clip(dev, kinds):
    """batch_size 1: the reference's util.sample_* return scalars for n = 1 (so do the mirrors); the batch-wide draws
    must still work, and the one clip is rendered like any other (flanger / chorus bit-exact, phaser 1e-5)."""
    from mod_extraction_amd import data_modules
    from oracle import lightning as ol
    torch.manual_seed(8); np.random.seed(8)
    n, sr = 22272, 44100
    bt = data_modules.SyntheticFxBatcher(1, n, sr, kinds, dev, audio_seed=3)
    params = bt.sample_params()
    dry, wet, mod, fxp = bt.render(params)
    assert dry.shape == wet.shape == (1, 1, n) and mod.shape == (1, n // 100)
    assert torch.isfinite(wet).all() and torch.isfinite(mod).all()
    if kinds[0] == "dry":
        assert torch.equal(dry, wet)
        return
    d_r, w_r, m_r = ol.synth_batch(params, bt.kinds, bt.src.cpu().numpy(), n, sr, {"flanger": 1.0, "chorus": 30.0},
                                   mod_override=mod.cpu().numpy())
    assert torch.equal(dry.cpu(), d_r)
    if kinds[0] == "phaser":
        assert float((wet.cpu() - w_r).abs().max()) < 1e-5
    else:
        assert torch.equal(wet.cpu(), w_r)


@pytest.mark.parametrize("n", [1, 2, 3, 63, 64, 65, 882])
def test_lfo_and_interpolation_at_tiny_lengths(dev, n):
    """K1 + util.linear_interpolate_last_dim on rows as short as one point, against the oracle: phase bookkeeping shapes
    bit-exact, cosine family 1e-5; interpolation (align_corners) bit-exact, including n_in = 1 / n_out = 1."""
    from mod_extraction_amd import modulations as am, util as autil
    from oracle import util as outil
    for shape in ("tri", "saw", "rsaw"):
        want = omod.make_mod_signal(n, 441.0, 2.7, 0.9, shape)
        got = am.make_mod_signal(n, 441.0, 2.7, 0.9, shape, device=dev).cpu()
        assert torch.equal(got, want), shape
    for shape in ("cos", "rect_cos", "inv_rect_cos", "sqr"):
        want = omod.make_mod_signal(n, 441.0, 2.7, 0.9, shape)
        got = am.make_mod_signal(n, 441.0, 2.7, 0.9, shape, device=dev).cpu()
        assert float((got - want).abs().max()) < 1e-5, shape
    x = torch.rand(3, n)
    for n_out in (1, 2, 5, n, 2 * n + 1, 345):
        if n == 1 and n_out != 1:
            continue                                   # the reference's interpolation needs two points to stretch
        want = outil.linear_interpolate_last_dim(x, n_out)
        got = autil.linear_interpolate_last_dim(x.to(dev), n_out).cpu()
        assert torch.equal(got, want), (n, n_out)


@pytest.mark.parametrize("n", [3, 4, 5, 9, 33])
def test_corner_kernels_on_short_rows(dev, n):
    """K9 (find_corners, smoothen, stretch_corners, validity filter) on rows a few frames long, bit-exact against the oracle."""
    from mod_extraction_amd import modulations as am
    torch.manual_seed(n)
    m = torch.rand(4, n)
    t_o, b_o = omod.find_corners(m)
    t_g, b_g = am.find_corners(m.to(dev))
    assert torch.equal(t_g.cpu(), t_o.float()) and torch.equal(b_g.cpu(), b_o.float())
    for k in (0, 2, n):
        if 1 < k <= n:
            got, want = am.smoothen(m.to(dev), k).cpu(), omod.smoothen(m, k)
            if k <= 8:                                  # the windows the reference's configs use (4, 8): same summation order
                assert torch.equal(got, want), k
            else:                                       # torch's mean switches to a blocked sum for long rows: 1 ulp apart
                assert float((got - want).abs().max()) < 2e-7, k
    st_o = omod.stretch_corners(m, max_n_corners=16, smooth_n_frames=0)
    st_g = am.stretch_corners(m.to(dev), max_n_corners=16, smooth_n_frames=0).cpu()
    assert torch.equal(st_g, st_o)
    assert am.find_valid_mod_sig_indices(m.to(dev)) == omod.find_valid_mod_sig_indices(m)


@pytest.mark.parametrize("cfg", [
    dict(fft_sizes=(512, 2048), hop_sizes=(128, 512), win_lengths=(512, 2048)),        # full-length windows, 75 % overlap
    dict(fft_sizes=(2048,), hop_sizes=(2048,), win_lengths=(1024,)),                   # no overlap at all (hop = n_fft)
    dict(fft_sizes=(1024, 1024, 512), hop_sizes=(7, 1000, 500), win_lengths=(33, 1024, 100)),   # odd hops / tiny windows
])
def test_mrstft_other_resolution_sets(dev, cfg):
    """MultiResolutionSTFTLoss with resolution sets other than auraloss's defaults (the kernels take fft sizes, hops and
    windows as arguments): value against the fp64 oracle, gradient as in tests/test_gpu_mrstft.py."""
    from mod_extraction_amd import mrstft
    torch.manual_seed(len(cfg["fft_sizes"]))
    B, T = 2, 9001
    t = torch.arange(T) / 44100.0
    y = (0.5 * torch.sin(2 * np.pi * 330.0 * t) + 0.2 * torch.rand(B, 1, T) - 0.1).clamp(-1, 1)
    x = (0.8 * y + 0.1 * torch.roll(y, 7, -1) + 0.05 * torch.randn(B, 1, T)).clamp(-1, 1)

    class MR64(olosses.MultiResolutionSTFTLoss):
        def _mag(self, v, n_fft, hop, win):
            s = torch.stft(v.reshape(-1, v.size(-1)), n_fft, hop, win, torch.hann_window(win, dtype=torch.float64),
                           return_complex=True)
            return torch.sqrt(torch.clamp(s.real ** 2 + s.imag ** 2, min=self.eps))
    x64 = x.double().requires_grad_(True)
    loss64 = MR64(**cfg)(x64, y.double())
    loss64.backward()
    x32 = x.clone().requires_grad_(True)
    olosses.MultiResolutionSTFTLoss(**cfg)(x32, y).backward()
    xd = x.to(dev).requires_grad_(True)
    loss_m = mrstft.MultiResolutionSTFTLoss(**cfg)(xd, y.to(dev))
    loss_m.backward()
    assert abs(float(loss_m.detach()) - float(loss64.detach())) < 1e-5 * abs(float(loss64.detach()))
    scale = x64.grad.abs().max()
    e_mine = float((xd.grad.cpu().double() - x64.grad).abs().max() / scale)
    e_oracle32 = float((x32.grad.double() - x64.grad).abs().max() / scale)
    # full-length windows make the log-magnitude term's 1 / |X| steeper still: fp32 itself (the oracle in fp32) is only good
    # to ~3e-3 there, so the bound is the fp32 oracle's own error against fp64, not an absolute figure
    assert e_mine < 1e-2 and e_mine < max(2.0 * e_oracle32, 1e-4), (e_mine, e_oracle32)


@pytest.mark.parametrize("B,T", [(1, 1), (1, 63), (3, 65), (2, 4097), (5, 1000)])
def test_effect_loss_sums_at_odd_sizes(dev, B, T):
    """mx_effect_loss_sums (L1 / MSE / ESR / DC of losses.py:14-67) on clips that do not fill a wavefront or a block, against
    the oracle's torch expressions; a silent target exercises the eps of the ESR / DC denominators."""
    from mod_extraction_amd import effect_losses
    torch.manual_seed(B * 1000 + T)
    y = torch.rand(B, 1, T) * 1.6 - 0.8
    y[0] = 0.0                                              # silent target clip: denominators are eps
    x = (0.7 * y + 0.1 * torch.randn(B, 1, T)).clamp(-1, 1)
    got = effect_losses.effect_loss_terms(x.to(dev), y.to(dev))
    want = {"l1": torch.nn.functional.l1_loss(x, y), "mse": torch.nn.functional.mse_loss(x, y),
            "esr": olosses.ESRLoss()(x, y), "dc": olosses.DCLoss()(x, y)}
    for k, v in want.items():
        a, b = float(got[k]), float(v)
        assert abs(a - b) <= 2e-5 * max(1.0, abs(b)), (k, a, b)
