#!/usr/bin/env python3
"""Entry point mirroring the reference's scripts/train.py: pick a YAML config and run `fit`.
Run from this directory (config paths are relative to scripts/, as in the reference):
    python train.py [../configs/train_lfo_interwoven_all.yml]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py <config>
"""
import logging
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mod_extraction_amd.cli import CustomLightningCLI  # noqa: E402

logging.basicConfig()
log = logging.getLogger(__name__)
log.setLevel(level=os.environ.get("LOGLEVEL", "INFO"))

if __name__ == "__main__":
    config = sys.argv[1] if len(sys.argv) > 1 else os.path.join("..", "configs", "train_lfo_interwoven_all.yml")
    CustomLightningCLI(args=["fit", "-c", config])
