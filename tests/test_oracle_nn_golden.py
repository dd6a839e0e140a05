"""CPU: the torch-module part of the oracle (CNN stack, LSTM-64, losses, step logic) reproduces the
golden vectors produced by the reference's OWN classes (tests/golden/make_golden_nn.py: real
models.py / losses.py / lightning.py under name-only third-party stubs; the 7 shipped LSTM-64 weights)."""
import os

import numpy as np
import pytest
import torch

from oracle import lightning as ol, losses as olosses, models as om, modulations as omod


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def lstm_from_golden(g, i):
    m = om.LSTMEffectModel(1, 1, 64, 1)
    sd = {k[len(f"w_{i}_"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(f"w_{i}_")}
    m.load_state_dict(sd, strict=True)
    return m


def test_lstm_with_shipped_weights(golden_dir):
    g = load(golden_dir, "lstm.npz")
    x, lat = torch.from_numpy(g["x"]), torch.from_numpy(g["latent"])
    T = x.size(-1) // 2
    for i in range(int(g["n_files"])):
        m = lstm_from_golden(g, i)
        with torch.no_grad():
            y1 = m(x[..., :T], lat[..., :T])
            m.detach_hidden()
            y2 = m(x[..., T:], lat[..., T:])
        assert torch.equal(torch.cat([y1, y2], -1), torch.from_numpy(g[f"y_{i}"])), str(g[f"name_{i}"])
        assert torch.equal(m.hidden[0], torch.from_numpy(g[f"h_{i}"]))
        assert torch.equal(m.hidden[1], torch.from_numpy(g[f"c_{i}"]))


def test_cnn_stack_matches_reference_class(golden_dir):
    g = load(golden_dir, "cnn_stack.npz")
    torch.manual_seed(int(g["seed"]))
    m = om.Spectral2DCNN(in_ch=2, n_samples=22272, sr=44100, n_fft=1024, hop_len=256, n_mels=64, kernel_size=(5, 13),
                         out_channels=[64] * 6, temp_dilations=[1, 1, 2, 4, 8, 16], pool_size=(2, 1), latent_dim=1,
                         freq_mask_amount=0.25, time_mask_amount=0.25, use_ln=True).eval()
    assert sum(p.numel() for p in m.parameters()) == int(g["n_params"])
    assert np.array_equal(m.state_dict()["cnn.1.weight"].numpy()[:4], g["w_first"])     # same seeded init
    want_keys = [str(k) for k in g["keys"]]
    assert [k for k in m.state_dict() if not k.startswith("spectrogram")] == want_keys
    with torch.no_grad():
        latent = torch.mean(m.cnn(torch.from_numpy(g["logmel"])), dim=-2)
        y = torch.sigmoid(m.output(latent))
    assert torch.equal(latent, torch.from_numpy(g["latent"]))
    assert torch.equal(y, torch.from_numpy(g["y"]))


def test_full_size_parameter_count():
    m = om.Spectral2DCNN(in_ch=2, n_mels=256, out_channels=[64] * 6, temp_dilations=[1, 1, 2, 4, 8, 16], pool_size=(2, 1))
    assert sum(p.numel() for p in m.parameters()) == 1340353          # 1.3 M parameters
    assert sum(p.numel() for p in om.LSTMEffectModel().parameters()) == 17473


def test_losses(golden_dir):
    g = load(golden_dir, "losses.npz")
    a, b = torch.from_numpy(g["a"]), torch.from_numpy(g["b"])
    wa, wb = torch.from_numpy(g["wa"]), torch.from_numpy(g["wb"])
    for name in ("l1", "fdl1", "sdl1", "mse"):
        assert float(olosses.get_loss_func_by_name(name)(a, b)) == float(g[name]), name
    for name in ("l1", "esr", "dc"):
        assert float(olosses.get_loss_func_by_name(name)(wa, wb)) == float(g["w_" + name]), name
    with pytest.raises(KeyError):
        olosses.get_loss_func_by_name("nope")


class _Preset(torch.nn.Module):
    def __init__(self, y):
        super().__init__()
        self.y = y

    def forward(self, x):
        return self.y.unsqueeze(1), None


@pytest.mark.parametrize("tag,kw", [("train", dict(model_smooth_n_frames=0, should_stretch=False)),
                                    ("eval4", dict(model_smooth_n_frames=4, should_stretch=False)),
                                    ("stretch", dict(model_smooth_n_frames=8, should_stretch=True, max_n_corners=16))])
def test_lfo_extraction_step_logic(golden_dir, tag, kw):
    g = load(golden_dir, "steps.npz")
    y_hat, mod = torch.from_numpy(g["lfo_y_hat"]), torch.from_numpy(g["lfo_mod"])
    loss_dict = {"l1": 1.0, "fdl1": 5.0, "sdl1": 10.0, "mse": 0.0}
    loss, terms, hat = ol.lfo_common_step(_Preset(y_hat), None, torch.zeros(5, 1, 10), mod, loss_dict, use_dry=False, **kw)
    assert float(loss) == float(g[f"lfo_{tag}_loss"])
    names = [str(n) for n in g[f"lfo_{tag}_names"]]
    assert names == ["val/l1", "val/fdl1", "val/sdl1", "val/mse", "val/loss"]
    for n, v in zip(names[:-1], g[f"lfo_{tag}_logged"][:-1]):
        assert float(terms[n.split("/")[1]]) == float(v), n
    assert np.array_equal(hat.detach().numpy(), g[f"lfo_{tag}_hat"])


def test_tbptt_step_logic(golden_dir):
    g = load(golden_dir, "steps.npz")
    em = om.LSTMEffectModel(1, 1, 64, 1)
    em.load_state_dict({k[len("tb_init_"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("tb_init_")})
    opt = torch.optim.AdamW(em.parameters(), lr=1e-4, betas=(0.8, 0.99))
    res = ol.tbptt_common_step(em, opt, torch.from_numpy(g["tb_dry"]), torch.from_numpy(g["tb_wet"]),
                               torch.from_numpy(g["tb_lfo"]), 256, 256, {"l1": 1.0, "esr": 0.0, "dc": 0.0})
    assert len(res["kept"]) == int(g["tb_kept"]) and res["steps"] == int(g["tb_steps"])
    assert np.array_equal(res["mod_sig_hat"].numpy(), g["tb_mod_sig_hat"])
    assert np.array_equal(res["wet_hat"].numpy(), g["tb_wet_hat"])
    assert float(res["loss"]) == float(g["tb_loss"])
    names = [str(n) for n in g["tb_names"]]
    assert names == ["train/l1", "train/esr", "train/dc", "train/loss"]
    for n, v in zip(names[:-1], g["tb_logged"][:-1]):
        assert float(res["terms"][n.split("/")[1]]) == float(v), n
    for k, v in em.state_dict().items():
        assert np.array_equal(v.numpy(), g[f"tb_final_{k}"]), k
