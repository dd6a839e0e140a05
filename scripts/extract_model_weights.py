#!/usr/bin/env python3
"""scripts/extract_model_weights.py of the reference with its hard-coded names as arguments: take the weights of ONE
sub-module (`effect_model` of a TBPTT run, `model` of an LFO-extraction run, ...) out of a training checkpoint and save
them as a plain state dict `<name>.pt` -- the file `lfo_model_weights_path` of configs/train_em_dry_wet.yml points at.
    python extract_model_weights.py <name> [--attr effect_model] [--dir ../models]
expects `<dir>/<name>.yml` (the run's config) and `<dir>/<name>.ckpt` (written by this package's `fit` or by Lightning).
"""
import argparse
import logging
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mod_extraction_amd.cli import CustomLightningCLI  # noqa: E402
from mod_extraction_amd.paths import MODELS_DIR  # noqa: E402

logging.basicConfig()
log = logging.getLogger(__name__)
log.setLevel(level=os.environ.get("LOGLEVEL", "INFO"))


def extract(model_dir: str, model_name: str, attr: str, device=None) -> str:
    config_path = os.path.join(model_dir, f"{model_name}.yml")
    ckpt_path = os.path.join(model_dir, f"{model_name}.ckpt")
    ckpt = torch.load(ckpt_path, map_location="cpu", weights_only=False)
    cli = CustomLightningCLI(args=["-c", config_path], trainer_defaults=CustomLightningCLI.trainer_defaults, run=False,
                             device=device, allow_missing_ckpt=True)
    assert hasattr(cli.model, attr), f"{type(cli.model).__name__} has no sub-module {attr!r}"
    module = getattr(cli.model, attr)
    tag = f"{attr}."
    state = {k[len(tag):]: v for k, v in ckpt["state_dict"].items() if k.startswith(tag)}
    module.load_state_dict(state)                         # strict: the checkpoint must cover the sub-module
    save_path = os.path.join(model_dir, f"{model_name}.pt")
    torch.save({k: v.detach().cpu() for k, v in module.state_dict().items()}, save_path)
    log.info("wrote %s (%d tensors)", save_path, len(state))
    return save_path


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("name")
    ap.add_argument("--attr", default="effect_model")
    ap.add_argument("--dir", default=MODELS_DIR)
    a = ap.parse_args()
    extract(a.dir, a.name, a.attr)
