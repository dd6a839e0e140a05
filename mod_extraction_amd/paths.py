"""Directory layout of the reference's mod_extraction/paths.py (repository root, configs/, data/, models/, out/).
The reference asserts that data/ and out/ exist at import time; here they are created on first use instead
(`ensure(path)`), so that importing the package never fails on a fresh checkout."""
import os

ROOT_DIR = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

CONFIGS_DIR = os.path.join(ROOT_DIR, "configs")
DATA_DIR = os.path.join(ROOT_DIR, "data")
MODELS_DIR = os.path.join(ROOT_DIR, "models")
OUT_DIR = os.path.join(ROOT_DIR, "out")


def ensure(path: str) -> str:
    os.makedirs(path, exist_ok=True)
    return path
