"""CPU: the YAML entry layer (mod_extraction_amd/cli.py) builds the same object graph from this repo's
configs and -- where /root/reference is mounted -- from the reference's own shipped configs."""
import os

import pytest
import torch

from mod_extraction_amd import cli, data_modules, lightning, models, optim

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
CPU = torch.device("cpu")


def build(path, cwd):
    old = os.getcwd()
    os.chdir(cwd)
    try:
        # the pretrained extractor blob eval_lfo.yml names is not part of the reference repository
        return cli.CustomLightningCLI(args=["fit", "-c", path], run=False, device=CPU, allow_missing_ckpt=True)
    finally:
        os.chdir(old)


def test_interwoven_config_object_graph():
    c = build("../configs/train_lfo_interwoven_all.yml", os.path.join(ROOT, "scripts"))
    assert isinstance(c.model, lightning.LFOExtraction) and isinstance(c.model.model, models.Spectral2DCNN)
    assert c.model.model.n_frames == 345 and c.model.model.in_ch == 2            # n_samples linked from data
    assert c.model.loss_dict == {"l1": 1.0, "fdl1": 5.0, "sdl1": 10.0, "mse": 0.0}
    assert isinstance(c.datamodule, data_modules.InterwovenDataModule) and c.datamodule.batch_size == 256
    assert c.datamodule.kinds == ("flanger", "chorus", "phaser")
    assert cli.resolve_class(c.optimizer_spec["class_path"]) is optim.FlatAdamW
    assert float(c.optimizer_spec["init_args"]["lr"]) == 1e-4 and c.optimizer_spec["init_args"]["betas"] == [0.8, 0.99]
    assert sum(p.numel() for p in c.model.parameters()) == 1340353
    assert c.trainer.max_epochs == 400 and c.trainer.num_sanity_val_steps == 2


@pytest.mark.parametrize("name,kind", [("train_lfo_phaser.yml", lightning.LFOExtraction),
                                       ("eval_lfo.yml", lightning.LFOExtraction),
                                       ("train_em_dry_wet.yml", lightning.TBPTTLFOEffectModeling)])
def test_other_configs(name, kind):
    c = build(os.path.join("..", "configs", name), os.path.join(ROOT, "scripts"))
    assert isinstance(c.model, kind)
    if kind is lightning.TBPTTLFOEffectModeling:
        assert isinstance(c.model.effect_model, models.LSTMEffectModel)
        assert all(not p.requires_grad for p in c.model.lfo_model.parameters())
        assert sum(p.numel() for p in c.model.parameters() if p.requires_grad) == 17473
        assert c.model.automatic_optimization is False
    if name == "eval_lfo.yml":
        assert c.model.model_smooth_n_frames == 4 and c.datamodule.batch_size == 16


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not mounted (GPU box)")
@pytest.mark.parametrize("name", ["train_lfo_phaser.yml", "train_lfo_interwoven_all.yml", "train_lfo_flanger.yml",
                                  "train_em_dry_wet.yml", "train_baseline_em_dry_wet.yml", "prototyping_lfo_dry_wet.yml",
                                  "eval_lfo.yml", "eval_lfo_combined.yml", "eval_lfo_distorted.yml", "eval_lfo_quasi.yml",
                                  "eval_lfo_rand.yml", "eval_lfo_unseen_audio.yml", "eval_em_unseen_effect.yml"])
def test_reference_configs_parse(name):
    """EVERY top-level YAML the reference ships (read in place, never copied) resolves to this package's classes."""
    cfg = cli.apply_links(cli.load_config(os.path.join(REF, "configs", name)))
    # the pretrained LFO-net .pt / .ckpt are large blobs absent from the mount
    if "lfo_model_weights_path" in cfg["model"]["init_args"]:
        cfg["model"]["init_args"]["lfo_model_weights_path"] = None
    model = cli.instantiate(cfg["model"])
    data = cli.instantiate(cfg["data"])
    assert type(model).__name__ == cfg["model"]["class_path"].rsplit(".", 1)[1]
    assert type(data).__name__ == cfg["data"]["class_path"].rsplit(".", 1)[1]
    inner = getattr(model, "model", None) or getattr(model, "lfo_model", None)
    assert isinstance(inner, (models.Spectral2DCNN, models.RandomLFO))
    if isinstance(inner, models.Spectral2DCNN):
        assert inner.n_frames == 345
    else:                                   # baseline_rand_lfo.yml: the values written in the YAML survive the links
        assert (inner.n_samples, inner.sr) == (345, 172.5)
    if "optimizer" in cfg:
        assert cli.resolve_class(cfg["optimizer"]["class_path"]) is optim.FlatAdamW


def test_lstm_state_dict_round_trip(golden_dir):
    """the shipped LSTM-64 state-dict keys load into the mirror with strict=True (on the CPU: no compute)."""
    import numpy as np
    g = np.load(os.path.join(golden_dir, "lstm.npz"))
    sd = {k[len("w_0_"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w_0_")}
    m = models.LSTMEffectModel()
    m.load_state_dict(sd, strict=True)
    assert list(m.state_dict().keys()) == ["lstm.weight_ih_l0", "lstm.weight_hh_l0", "lstm.bias_ih_l0",
                                           "lstm.bias_hh_l0", "fc.weight", "fc.bias"]


def test_missing_ckpt_path_raises():
    """Lightning raises for a ckpt_path that does not exist; so does this entry layer (no silent random weights)."""
    old = os.getcwd()
    os.chdir(os.path.join(ROOT, "scripts"))
    try:
        with pytest.raises(FileNotFoundError):
            cli.CustomLightningCLI(args=["validate", "-c", "../configs/eval_lfo.yml"], run=False, device=CPU)
    finally:
        os.chdir(old)


def test_fractional_batch_limits():
    from mod_extraction_amd import trainer
    assert trainer._limit(100, None) == 100 and trainer._limit(100, 7) == 7 and trainer._limit(5, 7) == 5
    assert trainer._limit(100, 0.1) == 10 and trainer._limit(100, 1.0) == 100 and trainer._limit(100, 2.0) == 2


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not mounted (GPU box)")
def test_reference_trained_and_model_configs_parse():
    """configs/trained/*.yml (the 27 experiment configs behind the paper's tables) and configs/models/*.yml, read in
    place: every object graph builds.  models/spectral_tcn.yml carries a stale `smooth_n_frames` argument that the
    reference's own SpectralTCN rejects; the mirror rejects it the same way."""
    import glob
    trained = sorted(glob.glob(os.path.join(REF, "configs", "trained", "*.yml")))
    assert len(trained) >= 20
    for path in trained:
        cfg = cli.apply_links(cli.load_config(path))
        ia = cfg["model"].get("init_args", {})
        if "lfo_model_weights_path" in ia:
            ia["lfo_model_weights_path"] = None
        model, data = cli.instantiate(cfg["model"]), cli.instantiate(cfg["data"])
        assert type(model).__name__ == cfg["model"]["class_path"].rsplit(".", 1)[1], path
        assert type(data).__name__ == cfg["data"]["class_path"].rsplit(".", 1)[1], path
    for path in sorted(glob.glob(os.path.join(REF, "configs", "models", "*.yml"))):
        cfg = cli.load_config(path)
        if os.path.basename(path) == "spectral_tcn.yml":
            with pytest.raises(TypeError):
                cli.instantiate(cfg)
            cfg["init_args"].pop("smooth_n_frames")
        obj = cli.instantiate(cfg)
        assert type(obj).__name__ == cfg["class_path"].rsplit(".", 1)[1], path


def test_cli_without_subcommand_builds_the_graph_only():
    """LightningCLI(args=["-c", cfg], run=False) as scripts/extract_model_weights.py uses it; a --ckpt_path on the command
    line overrides the config's; running without a subcommand is refused."""
    old = os.getcwd()
    os.chdir(os.path.join(ROOT, "scripts"))
    try:
        c = cli.CustomLightningCLI(args=["-c", "../configs/train_lfo_phaser.yml"], run=False, device=CPU,
                                   trainer_defaults=cli.CustomLightningCLI.trainer_defaults)
        assert c.subcommand is None and isinstance(c.model, lightning.LFOExtraction)
        with pytest.raises(AssertionError):
            cli.CustomLightningCLI(args=["-c", "../configs/train_lfo_phaser.yml"], run=True, device=CPU)
        with pytest.raises(FileNotFoundError):
            cli.CustomLightningCLI(args=["validate", "--config", "../configs/train_lfo_phaser.yml", "--ckpt_path",
                                         "/nonexistent/x.ckpt"], run=False, device=CPU)
    finally:
        os.chdir(old)


def test_links_fill_missing_values_and_keep_explicit_ones(tmp_path):
    """cli.py:71-103: data.n_samples / sr reach the nested model when its YAML leaves them out, and stay out of the way
    when it sets them."""
    import textwrap
    base = """
        data:
          class_path: mod_extraction.data_modules.FlangerCPUDataModule
          init_args: {batch_size: 2, n_samples: 22272, sr: 44100}
        model:
          class_path: mod_extraction.lightning.LFOExtraction
          init_args:
            use_dry: false
            model:
              class_path: mod_extraction.models.RandomLFO
              init_args: {%s}
    """
    for inner, want in (("n_samples: 87, sr: 172.5", (87, 172.5)), ("sr: 172.5", (22272, 172.5)), ("", None)):
        p = tmp_path / "c.yml"
        p.write_text(textwrap.dedent(base % inner))
        cfg = cli.apply_links(cli.load_config(str(p)))
        ia = cfg["model"]["init_args"]["model"]["init_args"] or {}
        if want is None:
            assert ia.get("n_samples") == 22272 and ia.get("sr") == 44100
        else:
            assert (ia["n_samples"], ia["sr"]) == want
