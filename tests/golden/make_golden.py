"""Generate the golden fixtures under tests/golden/ from the REAL reference.

Run only in the build container, where /root/reference exists:

    PYTHONPATH=/root/reference:/root/repo PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference's importable modules (mod_extraction.{util,modulations,fx}) are imported as they
are.  ``models.py`` / ``losses.py`` / ``lightning.py`` need third-party packages that are absent
(torchaudio, auraloss, pytorch_lightning, ...); they are imported with throw-away ``sys.modules``
stubs that provide *names only* -- no stub ever computes anything that ends up in a fixture (the
mel front end, MR-STFT and the Lightning trainer are not exercised here).
Fixtures are data only: inputs, parameters and the reference's outputs.
"""
import math
import os
import sys
import types

import numpy as np
import torch as tr

OUT = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
SHAPES = ["cos", "rect_cos", "inv_rect_cos", "tri", "saw", "rsaw", "sqr"]


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KB")


def gen_lfo(rmod):
    rng = np.random.default_rng(1234)
    rows = []
    for n, sr in ((882, 441.0), (345, 172.5), (2000, 44100.0)):
        for si, shape in enumerate(SHAPES):
            for exp in (1.0, 2.0, 0.7):
                f = float(np.exp(rng.uniform(np.log(0.5), np.log(3.0))))
                ph = float(rng.uniform(0, 2 * math.pi))
                y = rmod.make_mod_signal(n, sr, f, ph, shape, exp).numpy()
                rows.append((n, sr, f, ph, si, exp, y))
    save("lfo.npz",
         n=np.array([r[0] for r in rows]), sr=np.array([r[1] for r in rows]),
         freq=np.array([r[2] for r in rows]), phase=np.array([r[3] for r in rows]),
         shape=np.array([r[4] for r in rows]), exp=np.array([r[5] for r in rows]),
         **{f"y{i}": r[6] for i, r in enumerate(rows)})
    # phaser ground-truth LFO: crop of a longer cos LFO resampled to 882 points (datasets.py:442-452)
    crops = []
    for rate, start in ((0.5, 0), (0.93, 17000), (2.9, 15206), (1.7, 123)):
        proc_n = 88200 + int((44100 / rate) + 0.5)
        full = rmod.make_mod_signal(proc_n, 44100, rate, tr.pi / 2, "cos")
        from mod_extraction import util as rutil
        crop = rutil.linear_interpolate_last_dim(full[start:start + 88200], 882, align_corners=True).numpy()
        crops.append((rate, start, proc_n, crop))
    save("lfo_phaser_gt.npz", rate=np.array([c[0] for c in crops]), start=np.array([c[1] for c in crops]),
         proc_n=np.array([c[2] for c in crops]), y=np.stack([c[3] for c in crops]))


def gen_interp(rutil):
    tr.manual_seed(7)
    out = {}
    for i, (a, b) in enumerate(((882, 88200), (882, 345), (338, 86410), (345, 342), (7, 50), (50, 7))):
        x = tr.rand(2, a)
        y = rutil.linear_interpolate_last_dim(x, b, align_corners=True)
        if b > 4000:                      # keep the fixture small: store a strided view + the full sum
            out[f"y{i}_idx"] = np.arange(0, b, 37)
            out[f"y{i}"] = y.numpy()[:, ::37]
            out[f"y{i}_sum"] = y.double().sum(-1).numpy()
        else:
            out[f"y{i}"] = y.numpy()
        out[f"x{i}"] = x.numpy()
        out[f"n{i}"] = np.array([a, b])
    save("interp.npz", **out)


def gen_flanger(rfx, rmod):
    tr.manual_seed(11)
    N, B, sr = 2000, 4, 44100
    out = {}
    ci = 0
    for mm, ml in ((1.0, 10.0), (30.0, 10.0), (1.0, 4.0)):
        ref = rfx.MonoFlangerChorusModule(B, 1, N, sr, mm, ml)
        x = tr.rand(B, 1, N) * 2 - 1
        mod = tr.stack([rmod.make_mod_signal(N, sr, 20.0 + 7 * i, 0.9 * i, s)
                        for i, s in enumerate(["cos", "tri", "saw", "inv_rect_cos"])])
        out[f"x_{ci}"], out[f"mod_{ci}"] = x.numpy()[:, 0], mod.numpy()
        out[f"ms_{ci}"] = np.array([mm, ml])
        psets = [
            dict(feedback=tr.rand(B) * 0.7, min_delay_width=tr.rand(B), width=tr.rand(B) * 0.75 + 0.25,
                 depth=tr.rand(B) * 0.75 + 0.25, mix=tr.rand(B) * 0.75 + 0.25),       # tensor params
            dict(feedback=0.3, min_delay_width=0.37, width=0.81, depth=0.9, mix=0.63),  # python floats
            dict(feedback=0.0, min_delay_width=0.0, width=1.0, depth=1.0, mix=1.0),     # d reaches 0 / <1
            dict(feedback=0.69, min_delay_width=0.0, width=0.0, depth=1.0, mix=0.5),    # d == 0 always
            dict(feedback=tr.tensor([0.0, 0.5, 0.69, 0.2]), min_delay_width=tr.tensor([0.0, 0.01, 1.0, 0.02]),
                 width=tr.tensor([0.001, 0.0, 1.0, 0.5]), depth=tr.tensor([1.0, 0.0, 1.0, 0.5]),
                 mix=tr.tensor([0.0, 1.0, 1.0, 0.5])),                                 # mix in {0,1}, d<1
        ]
        for pi, p in enumerate(psets):
            y = ref(x, mod, **p)
            out[f"y_{ci}_{pi}"] = y.numpy()[:, 0]
            for k, v in p.items():
                out[f"p_{ci}_{pi}_{k}"] = v.numpy() if isinstance(v, tr.Tensor) else np.float64(v)
            out[f"p_{ci}_{pi}_is_tensor"] = np.array(isinstance(p["mix"], tr.Tensor))
        ci += 1
    out["n_cases"], out["n_psets"] = np.array(ci), np.array(5)
    save("flanger.npz", **out)

    # one full-length (2 s) clip pair through the real python loop (takes a few seconds):
    tr.manual_seed(12)
    N, B = 88200, 2
    ref = rfx.MonoFlangerChorusModule(B, 1, N, sr, 1.0, 10.0)
    x = tr.rand(B, 1, N) * 2 - 1
    freq, phase = [0.7, 2.6], [1.0, 4.0]
    lfo = tr.stack([rmod.make_mod_signal(882, 441.0, f, p, s) for f, p, s in zip(freq, phase, ["tri", "cos"])])
    from mod_extraction import util as rutil
    mod = rutil.linear_interpolate_last_dim(lfo, N)
    p = dict(feedback=tr.tensor([0.6, 0.1]), min_delay_width=tr.tensor([0.2, 0.9]),
             width=tr.tensor([0.9, 0.3]), depth=tr.tensor([0.8, 1.0]), mix=tr.tensor([0.7, 0.4]))
    y = ref(x, mod, **p)
    save("flanger_full.npz", seed=np.array(12), freq=np.array(freq), phase=np.array(phase),
         shape=np.array([3, 0]), lfo882=lfo.numpy(), x_sub=x.numpy()[:, 0, ::89], y_sub=y.numpy()[:, 0, ::89],
         y_sum=y.double().sum(-1).numpy()[:, 0], y_abs_sum=y.double().abs().sum(-1).numpy()[:, 0],
         **{f"p_{k}": v.numpy() for k, v in p.items()})


def gen_corners(rmod):
    tr.manual_seed(3)
    rng = np.random.default_rng(5)
    sigs = []
    for i in range(48):
        f = float(np.exp(rng.uniform(np.log(0.3), np.log(6.0))))
        ph = float(rng.uniform(0, 2 * math.pi))
        s = rmod.make_mod_signal(345, 172.5, f, ph, SHAPES[i % 7])
        if i % 3 == 0:
            s = (s * 0.7 + 0.1 + 0.02 * tr.randn(345)).clamp(0, 1)   # noisy: many corners (> max_n_corners)
        if i % 5 == 0:
            s = s * 0.5 + 0.2
        sigs.append(s)
    sigs.append(tr.full((345,), 0.5))                 # flat: zero corners
    sigs.append(tr.linspace(0, 1, 345))               # monotone: zero corners
    m = tr.stack(sigs)
    out = {"mod_sig": m.numpy()}
    for k in (0, 4, 8):
        ms = rmod.smoothen(m, k)
        top, bot = rmod.find_corners(ms)
        out[f"smooth_{k}"] = ms.numpy()
        out[f"top_{k}"], out[f"bot_{k}"] = top.numpy().astype(np.int8), bot.numpy().astype(np.int8)
        for mx in (16, 4):
            out[f"stretch_{k}_{mx}"] = rmod.stretch_corners(ms.clone(), mx, 0).numpy()
        out[f"valid_{k}"] = np.array(rmod.find_valid_mod_sig_indices(ms), dtype=np.int64)
    save("corners.npz", **out)


def gen_rng(rutil, rmod):
    """Seeded parameter streams (datasets.py:365-382 recipe; data_modules.py:419-443 order)."""
    tr.manual_seed(43)
    np.random.seed(43)
    items = []
    for _ in range(8):
        rate = rutil.sample_log_uniform(0.5, 3.0)
        phase = rutil.sample_uniform(0.0, 2 * math.pi)
        shape = rutil.choice(["cos", "rect_cos", "inv_rect_cos", "tri", "saw", "rsaw"])
        items.append((rate, phase, SHAPES.index(shape)))
    fb = rutil.sample_uniform(0.0, 0.7, n=8)
    mdw = rutil.sample_uniform(0.0, 1.0, n=8)
    width = rutil.sample_uniform(0.25, 1.0, n=8)
    depth = rutil.sample_uniform(0.25, 1.0, n=8)
    mix = rutil.sample_uniform(0.25, 1.0, n=8)
    save("param_stream.npz", rate=np.array([i[0] for i in items]), phase=np.array([i[1] for i in items]),
         shape=np.array([i[2] for i in items]), feedback=fb.numpy(), min_delay_width=mdw.numpy(),
         width=width.numpy(), depth=depth.numpy(), mix=mix.numpy())


def main():
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    from mod_extraction import fx as rfx, modulations as rmod, util as rutil
    gen_lfo(rmod)
    gen_interp(rutil)
    gen_flanger(rfx, rmod)
    gen_corners(rmod)
    gen_rng(rutil, rmod)
    try:
        from make_golden_nn import main as nn_main     # models / losses / lightning goldens
        nn_main()
    except ImportError:
        pass


if __name__ == "__main__":
    main()
