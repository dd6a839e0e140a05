#!/usr/bin/env python3
"""Entry point mirroring the reference's scripts/validate.py: run `validate` on a YAML config.
    python validate.py [../configs/eval_lfo.yml]
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
    config = sys.argv[1] if len(sys.argv) > 1 else os.path.join("..", "configs", "eval_lfo.yml")
    CustomLightningCLI(args=["validate", "-c", config])
