"""Golden vectors for MonoFlangerChorusModule with n_ch = 2 (fx.py:72-119: every channel has its own delay line, the
per-clip parameters are shared by a clip's channels, mod_sig is (bs, n) -- shared -- or (bs, n_ch, n)) -> flanger_stereo.npz.
Generated from the REAL reference module; only the vectors are committed.

    cd tests/golden && PYTHONDONTWRITEBYTECODE=1 python make_golden_stereo.py
"""
import os
import sys

import numpy as np
import torch as tr

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def main():
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    from mod_extraction import fx as rfx, modulations as rmod
    tr.manual_seed(21)
    N, B, C, sr = 1500, 3, 2, 44100
    out = {}
    for ci, (mm, ml) in enumerate(((1.0, 10.0), (30.0, 10.0))):
        ref = rfx.MonoFlangerChorusModule(B, C, N, sr, mm, ml)
        x = tr.rand(B, C, N) * 2 - 1
        mod_shared = tr.stack([rmod.make_mod_signal(N, sr, 15.0 + 9 * i, 0.7 * i, s) for i, s in enumerate(["cos", "tri", "saw"])])
        mod_per_ch = tr.stack([tr.stack([rmod.make_mod_signal(N, sr, 11.0 + 5 * i + 3 * c, 0.4 * i + c, s) for c in range(C)])
                               for i, s in enumerate(["rsaw", "cos", "rect_cos"])])
        p = dict(feedback=tr.rand(B) * 0.7, min_delay_width=tr.rand(B), width=tr.rand(B) * 0.75 + 0.25,
                 depth=tr.rand(B) * 0.75 + 0.25, mix=tr.rand(B) * 0.75 + 0.25)
        out[f"x_{ci}"], out[f"ms_{ci}"] = x.numpy(), np.array([mm, ml])
        out[f"mod_shared_{ci}"], out[f"mod_per_ch_{ci}"] = mod_shared.numpy(), mod_per_ch.numpy()
        for k, v in p.items():
            out[f"p_{ci}_{k}"] = v.numpy()
        out[f"y_shared_{ci}"] = ref(x, mod_shared, **p).numpy()
        out[f"y_per_ch_{ci}"] = ref(x, mod_per_ch, **p).numpy()
        out[f"y_float_{ci}"] = ref(x, mod_shared, feedback=0.4, min_delay_width=0.3, width=0.9, depth=0.8, mix=0.7).numpy()
    np.savez_compressed(os.path.join(HERE, "flanger_stereo.npz"), **out)
    print("wrote flanger_stereo.npz")


if __name__ == "__main__":
    main()
