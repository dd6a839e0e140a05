"""Golden vectors of the streaming effect model: the REAL `EffectModel` /
`EffectModelWrapper.do_forward_pass` of scripts/export_neutone_models.py (on-the-fly cosine LFO with a carried phase
and a stereo phase offset driving the shipped LSTM-64), imported in the build container with name-only stubs:

  neutone_sdk            -> `WaveformToWaveformBase` = nn.Module that stores `model`, `NeutoneParameter` / `save_neutone_model`
                            placeholders (packaging, no arithmetic)
  mod_extraction.paths   -> two directory constants (the real module asserts that data/ and out/ exist)
  torchaudio / auraloss / pytorch_lightning / plotting -> as in make_golden_nn.py

Four consecutive stereo buffers of different sizes through one model instance (hidden state and LFO phase carried),
weights = shipped file #0 of tests/golden/lstm.npz.  Run:  python tests/golden/make_golden_streaming.py
"""
import glob
import importlib.util
import os
import sys
import types

import numpy as np
import torch as tr
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
REF = "/root/reference"


def main():
    import make_golden_nn as g
    g.install_stubs()
    ns = types.ModuleType("neutone_sdk")

    class WaveformToWaveformBase(nn.Module):
        def __init__(self, model):
            super().__init__()
            self.model = model

    ns.WaveformToWaveformBase = WaveformToWaveformBase
    ns.NeutoneParameter = lambda *a, **k: None
    nsu = types.ModuleType("neutone_sdk.utils")
    nsu.save_neutone_model = lambda *a, **k: None
    ns.utils = nsu
    sys.modules["neutone_sdk"], sys.modules["neutone_sdk.utils"] = ns, nsu
    paths = types.ModuleType("mod_extraction.paths")
    paths.OUT_DIR, paths.MODELS_DIR = "/tmp", os.path.join(REF, "models")
    sys.path.insert(0, REF)
    import mod_extraction                                   # noqa: F401  (package first, then the stubbed submodule)
    sys.modules["mod_extraction.paths"] = paths
    spec = importlib.util.spec_from_file_location("ref_export_neutone", os.path.join(REF, "scripts", "export_neutone_models.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)

    files = sorted(glob.glob(os.path.join(REF, "models", "lstm_64__*.pt")))
    model = mod.EffectModel(weights_path=files[0])
    wrapper = mod.EffectModelWrapper(model)
    tr.manual_seed(31)
    sizes = [512, 2048, 300, 1024]
    params = {"lfo_rate": tr.tensor(0.33), "lfo_depth": tr.tensor(0.55), "lfo_stereo_phase_offset": tr.tensor(0.2)}
    out = {"sizes": np.array(sizes), "weights_index": np.array(0), "weights_name": np.array(os.path.basename(files[0]))}
    for k, v in params.items():
        out[f"p_{k}"] = v.numpy()
    model.model.clear_hidden()
    with tr.no_grad():
        for i, n in enumerate(sizes):
            x = tr.rand(2, n) * 1.2 - 0.6
            y = wrapper.do_forward_pass(x, params)
            out[f"x{i}"], out[f"y{i}"] = x.numpy(), y.numpy()
            out[f"phase{i}"] = np.array(float(model.prev_phase))
    path = os.path.join(HERE, "streaming.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
