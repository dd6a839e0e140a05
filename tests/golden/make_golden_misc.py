"""Golden vectors for the small host-driven pieces whose PRODUCT side had no reference-pinned test in round 1:
``make_rand_mod_signal`` (modulations.py:60-101, the RandomLFO baseline of models.py:19-69) and ``apply_tremolo``
(fx.py:13-22) -> rand_lfo_tremolo.npz; and the evaluation LFO variants ``make_quasi_periodic`` /
``make_concave_convex_mod_sig`` / ``make_combined_mod_sig`` (modulations.py:104-210) -> eval_lfo_variants.npz.
Generated from the REAL reference modules (importable as they are); only the vectors are committed.

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
    eval_variants(rmod)


# cases of the evaluation configs: eval_lfo_quasi.yml (l/r 0.1-0.3 style ranges), eval_lfo_combined.yml (shape lists),
# concave / convex distortion; every case = (seed, arguments) -> the reference's output under that host RNG stream
QUASI_CASES = [  # (seed, base shape, freq, phase, l_min, l_max, r_min, r_max, lr_split)
    (0, "cos", 2.3, 0.4, 0.1, 0.3, 0.1, 0.3, 0.5), (1, "cos", 2.3, 0.4, 0.1, 0.3, 0.1, 0.3, 0.5),
    (2, "tri", 1.1, 2.0, 0.2, 0.2, 0.2, 0.2, 0.5), (3, "saw", 2.9, 5.0, 0.0, 0.5, 0.0, 0.5, 0.3),
    (4, "rect_cos", 1.7, 1.0, 0.3, 0.4, 0.1, 0.2, 0.8), (5, "cos", 0.3, 0.0, 0.1, 0.3, 0.1, 0.3, 0.5),   # < 2 corners: unchanged
    (6, "inv_rect_cos", 2.6, 3.3, 0.05, 0.45, 0.05, 0.45, 0.5), (7, "rsaw", 0.9, 0.7, 0.25, 0.35, 0.25, 0.35, 0.5)]
COMBINED_CASES = [  # (seed, n, sr, freq, phase, shapes)
    (0, 882, 441.0, 2.3, 0.4, ["cos", "tri", "saw"]), (1, 882, 441.0, 2.3, 0.4, ["cos", "tri", "saw"]),
    (2, 882, 441.0, 0.7, 3.0, ["cos", "rect_cos", "inv_rect_cos", "tri", "saw", "rsaw"]),
    (3, 345, 172.5, 2.9, 6.0, ["tri", "rsaw"]), (4, 882, 441.0, 0.4, 1.0, ["cos", "saw"]),                # < 2 bottom corners
    (5, 1764, 882.0, 1.9, 2.2, ["cos", "rect_cos", "inv_rect_cos", "tri", "saw", "rsaw"])]
CONCAVE_CASES = [  # (seed, n, sr, freq, phase, concave_min, concave_max, convex_min, convex_max, concave_prob)
    (0, 882, 441.0, 1.7, 0.2, 0.2, 1.0, 1.0, 3.0, 0.5), (1, 882, 441.0, 2.9, 4.0, 0.2, 1.0, 1.0, 3.0, 0.5),
    (2, 345, 172.5, 0.6, 1.0, 0.5, 0.9, 1.5, 2.0, 0.2), (3, 882, 441.0, 1.0, 0.0, 0.2, 1.0, 1.0, 3.0, 1.0),
    (4, 882, 441.0, 2.2, 5.5, 0.2, 1.0, 1.0, 3.0, 0.0)]


def eval_variants(rmod):
    out = {}
    for i, (seed, shape, freq, phase, l0, l1, r0, r1, split) in enumerate(QUASI_CASES):
        base = rmod.make_mod_signal(882, 441.0, freq, phase, shape)
        tr.manual_seed(seed); np.random.seed(seed)
        out[f"quasi_{i}"] = rmod.make_quasi_periodic(base.clone(), l0, l1, r0, r1, split).numpy()
    for i, (seed, n, sr, freq, phase, shapes) in enumerate(COMBINED_CASES):
        tr.manual_seed(seed); np.random.seed(seed)
        out[f"combined_{i}"] = rmod.make_combined_mod_sig(n, sr, freq, phase, list(shapes)).numpy()
    for i, (seed, n, sr, freq, phase, a0, a1, b0, b1, prob) in enumerate(CONCAVE_CASES):
        tr.manual_seed(seed); np.random.seed(seed)
        out[f"concave_{i}"] = rmod.make_concave_convex_mod_sig(n, sr, freq, phase, a0, a1, b0, b1, prob).numpy()
    np.savez_compressed(os.path.join(HERE, "eval_lfo_variants.npz"), **out)
    print("wrote eval_lfo_variants.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
