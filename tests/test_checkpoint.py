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


class _FlatState:
    """the optimizer-state half of optim.FlatAdamW on the CPU (the step kernel itself needs a GPU)"""

    def __init__(self, params, lr=1e-4, betas=(0.8, 0.99), eps=1e-8, weight_decay=0.01):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        self.exp_avg, self.exp_avg_sq = torch.randn(n), torch.rand(n)
        self.step_count, self.lr, self.betas, self.eps, self.weight_decay = 37, lr, betas, eps, weight_decay


def test_checkpoint_is_loadable_by_the_reference_stack(tmp_path):
    """A .ckpt written here must be readable by what the reference uses: packaging.Version on the version field
    (Lightning's migrate_checkpoint), torch.optim.AdamW.load_state_dict on optimizer_states[0], and
    load_state_dict(strict=True) of the reference-equivalent modules under the LightningModule prefixes."""
    from packaging.version import Version
    from oracle import models as om
    torch.manual_seed(3)
    m = small_module()
    opt = _FlatState(m.parameters())
    keeper = trainer.CheckpointKeeper(str(tmp_path), "lfo_2dcnn", "synth", hyper_parameters={"config": {"a": 1}})
    keeper.update(m, opt, epoch=4, step=400, metrics={"val/loss": 0.25})
    blob = torch.load(tmp_path / "last.ckpt", weights_only=False)
    assert Version(blob["pytorch-lightning_version"]) >= Version("1.9.4")
    for key in ("epoch", "global_step", "state_dict", "optimizer_states", "lr_schedulers", "callbacks", "hyper_parameters"):
        assert key in blob, key
    assert blob["lr_schedulers"] == [] and blob["hyper_parameters"] == {"config": {"a": 1}}
    (cb_key, cb), = blob["callbacks"].items()
    assert cb_key.startswith("ModelCheckpoint") and cb["monitor"] == "val/loss" and float(cb["best_model_score"]) == 0.25
    assert cb["best_model_path"].endswith("lfo_2dcnn__synth__epoch_4_step_400.ckpt")
    # (a) the reference-equivalent extractor (oracle.models.Spectral2DCNN uses the reference's module tree) loads the
    #     "model."-prefixed weights strictly
    ref = om.Spectral2DCNN(in_ch=2, n_samples=22272, n_mels=64, out_channels=[64] * 6, temp_dilations=[1, 1, 2, 4, 8, 16],
                           pool_size=(2, 1))
    sd = {k[len("model."):]: v for k, v in blob["state_dict"].items() if k.startswith("model.")}
    ref.load_state_dict(sd, strict=True)
    # (b) torch.optim.AdamW takes the optimizer state as is, parameter by parameter
    t_opt = torch.optim.AdamW(ref.parameters(), lr=1.0)
    t_opt.load_state_dict(blob["optimizer_states"][0])
    g = t_opt.param_groups[0]
    assert g["lr"] == 1e-4 and tuple(g["betas"]) == (0.8, 0.99) and g["weight_decay"] == 0.01
    off = 0
    for p in ref.parameters():
        st = t_opt.state[p]
        assert float(st["step"]) == 37.0
        assert torch.equal(st["exp_avg"].reshape(-1), opt.exp_avg[off:off + p.numel()])
        assert torch.equal(st["exp_avg_sq"].reshape(-1), opt.exp_avg_sq[off:off + p.numel()])
        off += p.numel()
    assert off == opt.exp_avg.numel()
    # (c) resume: moments, step count, epoch and the best-score bookkeeping come back
    opt2 = _FlatState(small_module().parameters())
    opt2.exp_avg.zero_(); opt2.exp_avg_sq.zero_(); opt2.step_count = 0
    t2 = trainer.Trainer(max_epochs=10, log_fn=None, checkpoints=trainer.CheckpointKeeper(str(tmp_path / "x")))
    trainer.resume_from_checkpoint(str(tmp_path / "last.ckpt"), m, opt2, t2)
    assert opt2.step_count == 37 and torch.equal(opt2.exp_avg, opt.exp_avg) and torch.equal(opt2.exp_avg_sq, opt.exp_avg_sq)
    assert t2.start_epoch == 5 and t2.checkpoints.best == 0.25


def test_tbptt_checkpoint_prefixes(tmp_path):
    """effect_model. / lfo_model. prefixes of TBPTTLFOEffectModeling (lightning.py:237-246) load strictly into the
    reference-equivalent LSTM."""
    from oracle import models as om
    cnn = models.Spectral2DCNN(in_ch=2, n_samples=22272, n_mels=64, out_channels=[64] * 6,
                               temp_dilations=[1, 1, 2, 4, 8, 16], pool_size=(2, 1))
    tb = lightning.TBPTTLFOEffectModeling(1024, 1024, models.LSTMEffectModel(), lfo_model=cnn)
    trainer.save_checkpoint(str(tmp_path / "em.ckpt"), tb, None, 0, 83)
    sd = torch.load(tmp_path / "em.ckpt", weights_only=False)["state_dict"]
    assert {k.split(".")[0] for k in sd} == {"effect_model", "lfo_model"}
    ref = om.LSTMEffectModel()
    ref.load_state_dict({k[len("effect_model."):]: v for k, v in sd.items() if k.startswith("effect_model.")}, strict=True)


def test_resume_never_deletes_the_previous_runs_checkpoints(tmp_path):
    """`fit --ckpt_path run1/best.ckpt` continues in a NEW version directory; the first improving epoch of
    the resumed run must not remove run 1's best file (Lightning's ModelCheckpoint forgets best_k_models when dirpath
    changed).  The best SCORE carries over; within one directory the old best is still replaced."""
    torch.manual_seed(4)
    m = small_module()
    run1, run2 = tmp_path / "version_0" / "checkpoints", tmp_path / "version_1" / "checkpoints"
    k1 = trainer.CheckpointKeeper(str(run1), "lfo_2dcnn", "synth")
    k1.update(m, None, epoch=0, step=10, metrics={"val/loss": 0.5})
    best1 = run1 / "lfo_2dcnn__synth__epoch_0_step_10.ckpt"
    assert best1.exists()
    t2 = trainer.Trainer(max_epochs=3, log_fn=None, checkpoints=trainer.CheckpointKeeper(str(run2), "lfo_2dcnn", "synth"))
    trainer.resume_from_checkpoint(str(best1), m, None, t2)
    k2 = t2.checkpoints
    assert k2.best == 0.5 and k2.best_path is None               # the score carries over, the foreign path does not
    k2.update(m, None, epoch=1, step=20, metrics={"val/loss": 0.6})   # not better than run 1: only last.ckpt
    assert sorted(os.listdir(run2)) == ["last.ckpt"]
    k2.update(m, None, epoch=2, step=30, metrics={"val/loss": 0.4})   # improves: run 1's file must survive
    assert best1.exists() and (run1 / "last.ckpt").exists()
    assert sorted(os.listdir(run2)) == ["last.ckpt", "lfo_2dcnn__synth__epoch_2_step_30.ckpt"]
    k2.update(m, None, epoch=3, step=40, metrics={"val/loss": 0.3})   # own earlier best IS replaced
    assert sorted(os.listdir(run2)) == ["last.ckpt", "lfo_2dcnn__synth__epoch_3_step_40.ckpt"]
    # resuming INTO the same directory keeps managing that directory's best file
    k3 = trainer.CheckpointKeeper(str(run2), "lfo_2dcnn", "synth")
    k3.load_state(torch.load(run2 / "last.ckpt", weights_only=False)["callbacks"])
    assert k3.best_path is not None and os.path.basename(k3.best_path) == "lfo_2dcnn__synth__epoch_3_step_40.ckpt"


def test_resume_of_a_resume_keeps_the_best_score(tmp_path):
    """run 2 resumes run 1 into a new directory and never improves, so it has no best FILE of its own; its
    last.ckpt must still carry the score to beat, or run 3 (resuming run 2) would save its first epoch as 'best'."""
    torch.manual_seed(5)
    m = small_module()
    runs = [tmp_path / f"version_{i}" / "checkpoints" for i in range(3)]
    k1 = trainer.CheckpointKeeper(str(runs[0]), "lfo_2dcnn", "synth")
    k1.update(m, None, epoch=0, step=10, metrics={"val/loss": 0.5})
    t2 = trainer.Trainer(max_epochs=3, log_fn=None, checkpoints=trainer.CheckpointKeeper(str(runs[1]), "lfo_2dcnn", "synth"))
    trainer.resume_from_checkpoint(str(runs[0] / "last.ckpt"), m, None, t2)
    t2.checkpoints.update(m, None, epoch=1, step=20, metrics={"val/loss": 0.7})       # worse: only last.ckpt
    cb = torch.load(runs[1] / "last.ckpt", weights_only=False)["callbacks"]
    st = cb[trainer.CheckpointKeeper.STATE_KEY]
    assert float(st["best_model_score"]) == 0.5 and st["best_model_path"] == ""
    t3 = trainer.Trainer(max_epochs=3, log_fn=None, checkpoints=trainer.CheckpointKeeper(str(runs[2]), "lfo_2dcnn", "synth"))
    trainer.resume_from_checkpoint(str(runs[1] / "last.ckpt"), m, None, t3)
    k3 = t3.checkpoints
    assert k3.best == 0.5 and k3.best_path is None
    k3.update(m, None, epoch=2, step=30, metrics={"val/loss": 0.6})                   # still worse than run 1
    assert sorted(os.listdir(runs[2])) == ["last.ckpt"]
    # a checkpoint written by an older build (score None, finite kth_value) resumes through kth_value
    k4 = trainer.CheckpointKeeper(str(tmp_path / "v4"), "lfo_2dcnn", "synth")
    k4.load_state({trainer.CheckpointKeeper.STATE_KEY: {"best_model_score": None, "kth_value": torch.tensor(0.25), "dirpath": "/elsewhere"}})
    assert k4.best == 0.25 and k4.best_path is None
