"""CPU: the oracle restatements reproduce the golden vectors captured from the real reference
(tests/golden/make_golden.py).  Bit-exact unless stated."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import fx as ofx, modulations as omod, util as outil

SHAPES = ["cos", "rect_cos", "inv_rect_cos", "tri", "saw", "rsaw", "sqr"]


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_lfo_bit_exact(golden_dir):
    g = load(golden_dir, "lfo.npz")
    for i in range(len(g["n"])):
        y = omod.make_mod_signal(int(g["n"][i]), float(g["sr"][i]), float(g["freq"][i]), float(g["phase"][i]),
                                 SHAPES[int(g["shape"][i])], float(g["exp"][i])).numpy()
        assert np.array_equal(y, g[f"y{i}"], equal_nan=True), (i, SHAPES[int(g["shape"][i])])


def test_lfo_phaser_ground_truth(golden_dir):
    g = load(golden_dir, "lfo_phaser_gt.npz")
    for i in range(len(g["rate"])):
        full = omod.make_mod_signal(int(g["proc_n"][i]), 44100, float(g["rate"][i]), math.pi / 2, "cos")
        s = int(g["start"][i])
        crop = outil.linear_interpolate_last_dim(full[s:s + 88200], 882).numpy()
        assert np.array_equal(crop, g["y"][i])


def test_interp_bit_exact(golden_dir):
    g = load(golden_dir, "interp.npz")
    i = 0
    while f"x{i}" in g:
        a, b = g[f"n{i}"]
        y = outil.linear_interpolate_last_dim_np(g[f"x{i}"], int(b))
        if f"y{i}_idx" in g:
            assert np.array_equal(y[:, g[f"y{i}_idx"]], g[f"y{i}"])
            assert np.array_equal(y.astype(np.float64).sum(-1), g[f"y{i}_sum"])
        else:
            assert np.array_equal(y, g[f"y{i}"])
        i += 1
    assert i == 6


def _params(g, ci, pi):
    is_t = bool(g[f"p_{ci}_{pi}_is_tensor"])
    p = {}
    for k in ("feedback", "min_delay_width", "width", "depth", "mix"):
        v = g[f"p_{ci}_{pi}_{k}"]
        p[k] = torch.from_numpy(v.astype(np.float32)) if is_t else float(v)
    return p


def test_flanger_bit_exact(golden_dir):
    g = load(golden_dir, "flanger.npz")
    for ci in range(int(g["n_cases"])):
        x = torch.from_numpy(g[f"x_{ci}"]).unsqueeze(1)
        mod = torch.from_numpy(g[f"mod_{ci}"])
        mm, ml = g[f"ms_{ci}"]
        fl = ofx.MonoFlangerChorusModule(x.size(0), 1, x.size(-1), 44100, float(mm), float(ml))
        for pi in range(int(g["n_psets"])):
            y = fl(x, mod, **_params(g, ci, pi)).numpy()[:, 0]
            assert np.array_equal(y, g[f"y_{ci}_{pi}"]), (ci, pi)


def test_flanger_full_length(golden_dir):
    g = load(golden_dir, "flanger_full.npz")
    torch.manual_seed(int(g["seed"]))
    x = torch.rand(2, 1, 88200) * 2 - 1
    assert np.array_equal(x.numpy()[:, 0, ::89], g["x_sub"])
    lfo = torch.stack([omod.make_mod_signal(882, 441.0, float(f), float(p), SHAPES[int(s)])
                       for f, p, s in zip(g["freq"], g["phase"], g["shape"])])
    assert np.array_equal(lfo.numpy(), g["lfo882"])
    mod = outil.linear_interpolate_last_dim(lfo, 88200)
    p = {k: torch.from_numpy(g[f"p_{k}"]) for k in ("feedback", "min_delay_width", "width", "depth", "mix")}
    y = ofx.MonoFlangerChorusModule(2, 1, 88200, 44100, 1.0, 10.0)(x, mod, **p)
    assert np.array_equal(y.numpy()[:, 0, ::89], g["y_sub"])
    assert np.array_equal(y.double().sum(-1).numpy()[:, 0], g["y_sum"])


@pytest.mark.parametrize("k", [0, 4, 8])
def test_corner_bookkeeping_bit_exact(golden_dir, k):
    g = load(golden_dir, "corners.npz")
    m = torch.from_numpy(g["mod_sig"])
    ms = omod.smoothen(m, k)
    assert np.array_equal(ms.numpy(), g[f"smooth_{k}"])
    top, bot = omod.find_corners(ms)
    assert np.array_equal(top.numpy().astype(np.int8), g[f"top_{k}"])
    assert np.array_equal(bot.numpy().astype(np.int8), g[f"bot_{k}"])
    for mx in (16, 4):
        s = omod.stretch_corners(ms.clone(), mx, 0).numpy()
        assert np.array_equal(s, g[f"stretch_{k}_{mx}"], equal_nan=True)
    assert omod.find_valid_mod_sig_indices(ms) == g[f"valid_{k}"].tolist()


def test_param_stream(golden_dir):
    g = load(golden_dir, "param_stream.npz")
    torch.manual_seed(43)
    np.random.seed(43)
    for i in range(8):
        assert outil.sample_log_uniform(0.5, 3.0) == g["rate"][i]
        assert outil.sample_uniform(0.0, 2 * math.pi) == g["phase"][i]
        assert SHAPES.index(outil.choice(["cos", "rect_cos", "inv_rect_cos", "tri", "saw", "rsaw"])) == g["shape"][i]
    for name, (lo, hi) in (("feedback", (0.0, 0.7)), ("min_delay_width", (0.0, 1.0)), ("width", (0.25, 1.0)),
                           ("depth", (0.25, 1.0)), ("mix", (0.25, 1.0))):
        assert np.array_equal(outil.sample_uniform(lo, hi, n=8).numpy(), g[name])


def test_eval_lfo_variants_bit_exact(golden_dir):
    """(f) rank 2 -- make_quasi_periodic / make_combined_mod_sig / make_concave_convex_mod_sig (modulations.py:104-210):
    the oracle under the same host RNG seeds against vectors of the REAL functions (make_golden_misc.py)."""
    import sys
    sys.path.insert(0, golden_dir)
    import make_golden_misc as mg
    g = load(golden_dir, "eval_lfo_variants.npz")
    for i, (seed, shape, freq, phase, l0, l1, r0, r1, split) in enumerate(mg.QUASI_CASES):
        base = omod.make_mod_signal(882, 441.0, freq, phase, shape)
        torch.manual_seed(seed); np.random.seed(seed)
        y = omod.make_quasi_periodic(base.clone(), l0, l1, r0, r1, split).numpy()
        assert np.array_equal(y, g[f"quasi_{i}"]), ("quasi", i)
    for i, (seed, n, sr, freq, phase, shapes) in enumerate(mg.COMBINED_CASES):
        torch.manual_seed(seed); np.random.seed(seed)
        y = omod.make_combined_mod_sig(n, sr, freq, phase, list(shapes)).numpy()
        assert np.array_equal(y, g[f"combined_{i}"]), ("combined", i)
    for i, (seed, n, sr, freq, phase, a0, a1, b0, b1, prob) in enumerate(mg.CONCAVE_CASES):
        torch.manual_seed(seed); np.random.seed(seed)
        y = omod.make_concave_convex_mod_sig(n, sr, freq, phase, a0, a1, b0, b1, prob).numpy()
        assert np.array_equal(y, g[f"concave_{i}"]), ("concave", i)


def test_python_loop_flanger_equals_c_restatement():
    """oracle.fx.flanger_torch_loop (the reference's execution shape, used by bench.py's `reference_shaped` CPU leg)
    and the C restatement pinned by flanger.npz give the same bits."""
    torch.manual_seed(3)
    B, N = 3, 1500
    x, mod = torch.rand(B, 1, N) * 2 - 1, torch.rand(B, N)
    ps = [torch.rand(B) * 0.7, torch.rand(B), torch.rand(B) * 0.75 + 0.25, torch.rand(B) * 0.75 + 0.25, torch.rand(B) * 0.75 + 0.25]
    for m_min in (44, 1323):
        y = ofx.flanger_torch_loop(x, mod, m_min, 441, *ps)
        ref = ofx.MonoFlangerChorusModule(B, 1, N, 44100.0, m_min / 44.1, 10.0)
        assert ref.max_min_delay_samples == m_min
        assert torch.equal(y, ref(x, mod, *ps))


def test_flanger_two_channels_bit_exact(golden_dir):
    """MonoFlangerChorusModule with n_ch = 2 against vectors of the REAL module (tests/golden/make_golden_stereo.py): shared
    and per-channel mod_sig, tensor and float parameters -- every channel its own delay line, bit for bit."""
    import torch
    from oracle import fx as ofx
    g = np.load(os.path.join(golden_dir, "flanger_stereo.npz"))
    for ci in range(2):
        x = torch.from_numpy(g[f"x_{ci}"])
        mm, ml = (float(v) for v in g[f"ms_{ci}"])
        mod = ofx.MonoFlangerChorusModule(x.size(0), 2, x.size(-1), 44100, mm, ml)
        p = {k: torch.from_numpy(g[f"p_{ci}_{k}"]) for k in ("feedback", "min_delay_width", "width", "depth", "mix")}
        assert np.array_equal(mod(x, torch.from_numpy(g[f"mod_shared_{ci}"]), **p).numpy(), g[f"y_shared_{ci}"])
        assert np.array_equal(mod(x, torch.from_numpy(g[f"mod_per_ch_{ci}"]), **p).numpy(), g[f"y_per_ch_{ci}"])
        y = mod(x, torch.from_numpy(g[f"mod_shared_{ci}"]), feedback=0.4, min_delay_width=0.3, width=0.9, depth=0.8, mix=0.7)
        assert np.array_equal(y.numpy(), g[f"y_float_{ci}"])
