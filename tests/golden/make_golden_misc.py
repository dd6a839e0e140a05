"""Golden vectors for the small host-driven pieces whose PRODUCT side had no reference-pinned test in round 1:
``make_rand_mod_signal`` (modulations.py:60-101, the RandomLFO baseline of models.py:19-69) and ``apply_tremolo``
(fx.py:13-22).  Generated from the REAL reference modules (importable as they are); only the vectors are committed.

    cd tests/golden && PYTHONDONTWRITEBYTECODE=1 python make_golden_misc.py
"""
import math
import os
import sys

import numpy as np
import torch as tr

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
SHAPES = ["cos", "rect_cos", "inv_rect_cos", "tri", "saw", "rsaw"]


def main():
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    from mod_extraction import fx as rfx, modulations as rmod
    out = {}
    # (a) fully random rows (configs/models/baseline_egfx.yml: n 345 @ 172.5 Hz, rate 0.5-3)
    tr.manual_seed(7); np.random.seed(7)
    out["rand_a"] = rmod.make_rand_mod_signal(6, 345, 172.5, 0.5, 3.0).numpy()
    # (b) ground truth with errors (configs/models/baseline_rand_lfo.yml: phase_error 0.5, freq_error 0.25)
    tr.manual_seed(8); np.random.seed(8)
    phase_gt = tr.tensor([0.1, 1.5, 3.0, 4.4, 5.9, 6.2], dtype=tr.float32)
    freq_gt = tr.tensor([0.5, 0.9, 1.7, 2.2, 2.9, 3.0], dtype=tr.float32)
    shapes_gt = ["cos", "tri", "rect_cos", "inv_rect_cos", "saw", "rsaw"]
    out["gt_phase"], out["gt_freq"] = phase_gt.numpy().copy(), freq_gt.numpy().copy()
    out["gt_shape"] = np.array([SHAPES.index(s) for s in shapes_gt])
    out["rand_b"] = rmod.make_rand_mod_signal(6, 345, 172.5, 0.5, 3.0, shapes_gt, None, phase_gt.clone(), 0.5,
                                              freq_gt.clone(), 0.25).numpy()
    # (c) ground truth, no error: deterministic
    out["rand_c"] = rmod.make_rand_mod_signal(6, 345, 172.5, 0.5, 3.0, shapes_gt, None, phase_gt.clone(), 0.0,
                                              freq_gt.clone(), 0.0).numpy()
    # (d) random shapes from a restricted list with gt frequency only
    tr.manual_seed(9); np.random.seed(9)
    out["rand_d"] = rmod.make_rand_mod_signal(6, 345, 172.5, 0.5, 3.0, None, ["tri", "saw"], None, 0.5,
                                              freq_gt.clone(), 0.1).numpy()
    # apply_tremolo (fx.py:13-22)
    tr.manual_seed(10)
    x = tr.rand(3, 2, 500) * 2 - 1
    mod = tr.stack([rmod.make_mod_signal(500, 441.0, f, 0.3, s) for f, s in ((1.0, "cos"), (2.5, "tri"), (0.7, "saw"))])
    out["trem_x"], out["trem_mod"] = x.numpy(), mod.numpy()
    out["trem_y_07"] = rfx.apply_tremolo(x, mod, 0.7).numpy()
    out["trem_y_10"] = rfx.apply_tremolo(x, mod.unsqueeze(1).expand(-1, 2, -1), 1.0).numpy()
    out["trem_y_00"] = rfx.apply_tremolo(x, mod, 0.0).numpy()
    np.savez_compressed(os.path.join(HERE, "rand_lfo_tremolo.npz"), **out)
    print("wrote rand_lfo_tremolo.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
