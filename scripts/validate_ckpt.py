#!/usr/bin/env python3
"""scripts/validate_ckpt.py of the reference with its hard-coded name as an argument: `validate` a run's config against
the checkpoint stored next to it.
    python validate_ckpt.py <name> [--dir ../models]
expects `<dir>/<name>.yml` and `<dir>/<name>.ckpt`; a `ckpt_path` inside the config must name that same file.
"""
import argparse
import logging
import os
import sys

import yaml

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mod_extraction_amd.cli import CustomLightningCLI  # noqa: E402
from mod_extraction_amd.paths import MODELS_DIR  # noqa: E402

logging.basicConfig()
log = logging.getLogger(__name__)
log.setLevel(level=os.environ.get("LOGLEVEL", "INFO"))

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("name")
    ap.add_argument("--dir", default=MODELS_DIR)
    a = ap.parse_args()
    config_path = os.path.join(a.dir, f"{a.name}.yml")
    ckpt_path = os.path.join(a.dir, f"{a.name}.ckpt")
    with open(config_path, "r") as in_f:
        config = yaml.safe_load(in_f)
    if config.get("ckpt_path"):
        assert os.path.abspath(config["ckpt_path"]) == os.path.abspath(ckpt_path)
    CustomLightningCLI(args=["validate", "--config", config_path, "--ckpt_path", ckpt_path],
                       trainer_defaults=CustomLightningCLI.trainer_defaults)
