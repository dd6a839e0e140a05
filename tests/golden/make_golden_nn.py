"""Golden fixtures for the torch-module part of the path, produced by the REAL reference classes
(``mod_extraction.models`` / ``losses`` / ``lightning``), imported in the build container with
name-only stubs for the third-party packages that are absent from this image:

  torchaudio.transforms   -> placeholder classes (constructed by Spectral2DCNN.__init__, never called:
                             the fixtures drive ``model.cnn`` / ``model.output`` directly)
  auraloss.freq           -> placeholder (only the ``mrstft`` branch of the loss factory touches it)
  pytorch_lightning       -> a LightningModule base with ``log`` / ``optimizers`` / ``manual_backward``
                             plumbing (no arithmetic)
  mod_extraction.plotting -> two no-op functions (the real module needs librosa / matplotlib)

No stub computes anything that ends up in a fixture.  Run through tests/golden/make_golden.py.
"""
import glob
import os
import sys
import types

import numpy as np
import torch as tr
from torch import nn

OUT = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KB")


def install_stubs():
    class _Placeholder(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

        def forward(self, *a, **k):
            raise RuntimeError("third-party stub called: fixtures must not depend on it")

    ta = types.ModuleType("torchaudio")
    tat = types.ModuleType("torchaudio.transforms")
    for n in ("Spectrogram", "MelSpectrogram", "FrequencyMasking", "TimeMasking"):
        setattr(tat, n, _Placeholder)
    ta.transforms = tat
    sys.modules["torchaudio"], sys.modules["torchaudio.transforms"] = ta, tat

    au = types.ModuleType("auraloss")
    auf = types.ModuleType("auraloss.freq")
    auf.MultiResolutionSTFTLoss = _Placeholder
    au.freq = auf
    sys.modules["auraloss"], sys.modules["auraloss.freq"] = au, auf

    class LightningModule(nn.Module):
        def __init__(self):
            super().__init__()
            self._logged, self._opt = [], None
            self.automatic_optimization = True

        def log(self, name, value, **kw):
            self._logged.append((name, float(value)))

        def optimizers(self):
            return self._opt

        def manual_backward(self, loss):
            loss.backward()

    pl = types.ModuleType("pytorch_lightning")
    pl.LightningModule = LightningModule
    sys.modules["pytorch_lightning"] = pl

    plot = types.ModuleType("mod_extraction.plotting")
    plot.plot_spectrogram = lambda *a, **k: None
    plot.plot_mod_sig = lambda *a, **k: None
    sys.modules["mod_extraction.plotting"] = plot


def gen_lstm(rmodels):
    """LSTMEffectModel (models.py:311-339) with each of the 7 shipped weight files: two consecutive
    calls (hidden carried), outputs and final hidden state."""
    files = sorted(glob.glob(os.path.join(REF, "models", "lstm_64__*.pt")))
    assert len(files) == 7
    tr.manual_seed(21)
    T = 192
    x = tr.rand(2, 1, 2 * T) * 1.6 - 0.8
    lat = tr.rand(2, 1, 2 * T)
    out = {"x": x.numpy(), "latent": lat.numpy(), "n_files": np.array(len(files))}
    for i, f in enumerate(files):
        sd = tr.load(f, map_location="cpu")
        m = rmodels.LSTMEffectModel(in_ch=1, out_ch=1, n_hidden=64, latent_dim=1)
        m.load_state_dict(sd, strict=True)
        m.clear_hidden()
        with tr.no_grad():
            y1 = m(x[..., :T], lat[..., :T])
            m.detach_hidden()
            y2 = m(x[..., T:], lat[..., T:])
        out[f"name_{i}"] = np.array(os.path.basename(f))
        for k, v in sd.items():
            out[f"w_{i}_{k}"] = v.numpy()
        out[f"y_{i}"] = tr.cat([y1, y2], dim=-1).numpy()
        out[f"h_{i}"], out[f"c_{i}"] = m.hidden[0].numpy(), m.hidden[1].numpy()
    save("lstm.npz", **out)


def gen_cnn(rmodels):
    """Spectral2DCNN.cnn + mean + output + sigmoid (models.py:183-195,209-215) on a given log-mel tensor;
    weights come from the seeded default initialisation (same constructor order as the oracle)."""
    tr.manual_seed(1234)
    m = rmodels.Spectral2DCNN(in_ch=2, n_samples=22272, sr=44100, n_fft=1024, hop_len=256, n_mels=64,
                              kernel_size=(5, 13), out_channels=[64] * 6, temp_dilations=[1, 1, 2, 4, 8, 16],
                              pool_size=(2, 1), latent_dim=1, freq_mask_amount=0.25, time_mask_amount=0.25, use_ln=True)
    m.eval()
    logmel = tr.randn(1, 2, 64, 88) * 3.0 - 5.0
    with tr.no_grad():
        h = m.cnn(logmel)
        latent = tr.mean(h, dim=-2)
        y = tr.sigmoid(m.output(latent))
    keys = [k for k in m.state_dict().keys() if not k.startswith("spectrogram") and "masking" not in k]
    save("cnn_stack.npz", seed=np.array(1234), logmel=logmel.numpy(), y=y.numpy(), latent=latent.numpy(),
         keys=np.array(keys), w_first=m.state_dict()["cnn.1.weight"].numpy()[:4],
         n_params=np.array(sum(p.numel() for p in m.parameters())))


def gen_losses(rlosses):
    tr.manual_seed(5)
    a, b = tr.rand(6, 345), tr.rand(6, 345)
    wa, wb = tr.rand(4, 1, 3000) * 2 - 1, tr.rand(4, 1, 3000) * 2 - 1
    out = {"a": a.numpy(), "b": b.numpy(), "wa": wa.numpy(), "wb": wb.numpy()}
    for name in ("l1", "fdl1", "sdl1", "mse"):
        out[name] = rlosses.get_loss_func_by_name(name)(a, b).numpy()
    for name in ("l1", "esr", "dc"):
        out["w_" + name] = rlosses.get_loss_func_by_name(name)(wa, wb).numpy()
    save("losses.npz", **out)


def gen_steps(rlight, rmodels, rmod):
    """LFOExtraction.common_step (lightning.py:96-158) with a stand-in extractor that returns a preset
    mod_sig_hat, and TBPTTLFOEffectModeling.common_step (lightning.py:302-419) with ground-truth LFOs
    (lfo_model=None) on a small batch: per-step bookkeeping, logged losses, weights after training."""
    class Preset(nn.Module):
        def __init__(self, y):
            super().__init__()
            self.y = nn.Parameter(y)

        def forward(self, x):
            return self.y.unsqueeze(1), None

    tr.manual_seed(8)
    B = 5
    y_hat = tr.rand(B, 345)
    mod = tr.stack([rmod.make_mod_signal(882, 441.0, 0.7 + 0.4 * i, 0.5 * i, s)
                    for i, s in enumerate(["cos", "tri", "saw", "rsaw", "rect_cos"])])
    out = {"lfo_y_hat": y_hat.numpy(), "lfo_mod": mod.numpy()}
    for tag, kw in (("train", dict(model_smooth_n_frames=0, should_stretch=False)),
                    ("eval4", dict(model_smooth_n_frames=4, should_stretch=False)),
                    ("stretch", dict(model_smooth_n_frames=8, should_stretch=True, max_n_corners=16))):
        mdl = rlight.LFOExtraction(Preset(y_hat.clone()), use_dry=False,
                                   loss_dict={"l1": 1.0, "fdl1": 5.0, "sdl1": 10.0, "mse": 0.0}, **kw)
        loss, dd, _ = mdl.common_step((None, tr.zeros(B, 1, 10), mod.clone(), None), is_training=False)
        out[f"lfo_{tag}_loss"] = loss.detach().numpy()
        out[f"lfo_{tag}_logged"] = np.array([v for _, v in mdl._logged])
        out[f"lfo_{tag}_names"] = np.array([n for n, _ in mdl._logged])
        out[f"lfo_{tag}_target"] = dd["mod_sig"].numpy()
        out[f"lfo_{tag}_hat"] = dd["mod_sig_hat"].numpy()

    # TBPTT with ground-truth LFOs
    tr.manual_seed(9)
    B, n, W, S = 4, 3000, 256, 256
    dry = tr.rand(B, 1, n) * 1.6 - 0.8
    wet = (0.7 * dry + 0.2 * tr.roll(dry, 3, -1)).clamp(-1, 1)
    lfo = tr.stack([rmod.make_mod_signal(345, 172.5, f, p, s) for f, p, s in
                    ((1.1, 0.3, "cos"), (2.0, 1.0, "tri"), (0.2, 0.0, "cos"), (1.6, 2.0, "rect_cos"))])
    em = rmodels.LSTMEffectModel(1, 1, 64, 1)
    init = {k: v.clone() for k, v in em.state_dict().items()}
    mdl = rlight.TBPTTLFOEffectModeling(W, S, em, lfo_model=None, model_smooth_n_frames=8, should_stretch=True,
                                        max_n_corners=16, stretch_smooth_n_frames=0, discard_invalid_lfos=True,
                                        loss_dict={"l1": 1.0, "esr": 0.0, "dc": 0.0})
    mdl._opt = tr.optim.AdamW(em.parameters(), lr=1e-4, betas=(0.8, 0.99))
    loss, dd, _ = mdl.common_step((dry, wet, lfo.clone(), None), is_training=True)
    out.update(tb_dry=dry.numpy(), tb_wet=wet.numpy(), tb_lfo=lfo.numpy(), tb_loss=loss.detach().numpy(),
               tb_logged=np.array([v for _, v in mdl._logged]), tb_names=np.array([n_ for n_, _ in mdl._logged]),
               tb_wet_hat=dd["wet_hat"].numpy(), tb_mod_sig_hat=dd["mod_sig_hat"].numpy(),
               tb_kept=np.array(dd["dry"].shape[0]), tb_steps=np.array(len(mdl._opt.state_dict()["state"]) and
                                                                      int(list(mdl._opt.state_dict()["state"].values())[0]["step"])))
    for k, v in init.items():
        out[f"tb_init_{k}"] = v.numpy()
    for k, v in em.state_dict().items():
        out[f"tb_final_{k}"] = v.numpy()
    save("steps.npz", **out)


def main():
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    # paths.py asserts data/ and out/ exist next to the package; lightning.py does not import it
    install_stubs()
    from mod_extraction import models as rmodels, losses as rlosses, lightning as rlight, modulations as rmod
    gen_lstm(rmodels)
    gen_cnn(rmodels)
    gen_losses(rlosses)
    gen_steps(rlight, rmodels, rmod)


if __name__ == "__main__":
    main()
