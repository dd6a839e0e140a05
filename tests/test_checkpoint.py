"""CPU: Lightning-compatible checkpoint layout (state_dict prefixes, naming rule, best + last) and the
bare-weights extraction / re-loading paths of cli.load_weights."""
import os

import torch

from mod_extraction_amd import cli, lightning, models, trainer


def small_module():
    cnn = models.Spectral2DCNN(in_ch=2, n_samples=22272, n_mels=64, out_channels=[64] * 6,
                               temp_dilations=[1, 1, 2, 4, 8, 16], pool_size=(2, 1))
    return lightning.LFOExtraction(cnn, loss_dict={"l1": 1.0})


def test_checkpoint_round_trip(tmp_path):
    torch.manual_seed(0)
    m = small_module()
    keeper = trainer.CheckpointKeeper(str(tmp_path), "lfo_2dcnn", "synth")
    keeper.update(m, None, epoch=0, step=80, metrics={"val/loss": 0.5})
    keeper.update(m, None, epoch=1, step=160, metrics={"val/loss": 0.7})      # worse: only last.ckpt changes
    keeper.update(m, None, epoch=2, step=240, metrics={"val/loss": 0.3})      # better: replaces the best file
    files = sorted(os.listdir(tmp_path))
    assert files == ["last.ckpt", "lfo_2dcnn__synth__epoch_2_step_240.ckpt"]
    blob = torch.load(tmp_path / "last.ckpt")
    assert blob["epoch"] == 2 and blob["global_step"] == 240
    assert all(k.startswith("model.") for k in blob["state_dict"])           # LightningModule attribute prefix
    assert "model.cnn.1.weight" in blob["state_dict"] and "model.spectrogram.mel_scale.fb" in blob["state_dict"]
    # (a) load the full checkpoint into a fresh module
    torch.manual_seed(1)
    m2 = small_module()
    assert not torch.equal(m2.model.cnn[1].weight, m.model.cnn[1].weight)
    cli.load_weights(m2, str(tmp_path / "last.ckpt"))
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    # (b) extract the bare extractor weights (extract_model_weights.py) and load them as an lfo_model .pt
    bare = trainer.extract_model_weights(str(tmp_path / "last.ckpt"), str(tmp_path / "lfo.pt"), prefix="model.")
    assert "cnn.1.weight" in bare and not any(k.startswith("model.") for k in bare)
    cnn = models.Spectral2DCNN(in_ch=2, n_samples=22272, n_mels=64, out_channels=[64] * 6,
                               temp_dilations=[1, 1, 2, 4, 8, 16], pool_size=(2, 1))
    tb = lightning.TBPTTLFOEffectModeling(1024, 1024, models.LSTMEffectModel(), lfo_model=cnn,
                                          lfo_model_weights_path=str(tmp_path / "lfo.pt"))
    assert torch.equal(tb.lfo_model.cnn[1].weight, m.model.cnn[1].weight)
    # (c) a bare .pt loads into the wrapping module through cli.load_weights (prefix is found)
    torch.manual_seed(2)
    m3 = small_module()
    cli.load_weights(m3, str(tmp_path / "lfo.pt"))
    assert torch.equal(m3.model.output.weight, m.model.output.weight)
