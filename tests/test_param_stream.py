"""CPU: the PRODUCT's host-side sampling recipe (mod_extraction_amd.util RNG helpers, SyntheticFxBatcher.sample_params
in the reference's RNG order) against the parameter stream captured from the real reference
(tests/golden/make_golden.py -> param_stream.npz: datasets.py:365-398 per item, then data_modules.py:419-458 per
batch), and the oracle's restatements of make_rand_mod_signal / apply_tremolo against vectors from the real
modulations.py:60-101 / fx.py:13-22 (tests/golden/make_golden_misc.py)."""
import math
import os

import numpy as np
import pytest
import torch

SHAPES = ["cos", "rect_cos", "inv_rect_cos", "tri", "saw", "rsaw"]


def test_product_rng_helpers_reproduce_the_reference_stream(golden_dir):
    from mod_extraction_amd import util
    g = np.load(os.path.join(golden_dir, "param_stream.npz"))
    torch.manual_seed(43)
    np.random.seed(43)
    for i in range(8):
        assert util.sample_log_uniform(0.5, 3.0) == g["rate"][i]
        assert util.sample_uniform(0.0, 2 * math.pi) == g["phase"][i]
        assert SHAPES.index(util.choice(list(SHAPES))) == g["shape"][i]
    for name, (lo, hi) in (("feedback", (0.0, 0.7)), ("min_delay_width", (0.0, 1.0)), ("width", (0.25, 1.0)),
                           ("depth", (0.25, 1.0)), ("mix", (0.25, 1.0))):
        assert np.array_equal(util.sample_uniform(lo, hi, n=8).numpy(), g[name])
    assert util.sample_log_uniform(2.0, 2.0) == 2.0 and util.randint(3, 4) == 3


def test_batcher_reference_order_reproduces_the_flanger_datamodule_stream(golden_dir):
    """FlangerCPUDataModule with num_workers = 0: 8 items drawn one by one (rate, phase, shape), then the five
    (B,) effect-parameter draws -- value for value what the reference's generators yield under the same seeds."""
    from mod_extraction_amd.data_modules import SyntheticFxBatcher
    g = np.load(os.path.join(golden_dir, "param_stream.npz"))
    torch.manual_seed(43)
    np.random.seed(43)
    b = SyntheticFxBatcher(8, 88200, 44100, ("flanger",), torch.device("cpu"), rng_order="reference")
    p = b.sample_params()
    assert np.array_equal(p["rate_hz"].numpy(), g["rate"].astype(np.float32))
    assert np.array_equal(p["phase"].numpy(), g["phase"].astype(np.float32))
    assert [SHAPES.index(s) for s in p["shape"]] == g["shape"].tolist()
    for name in ("feedback", "min_delay_width", "width", "depth", "mix"):
        assert np.array_equal(p[name].numpy(), g[name]), name
    assert float(p["exp"][0]) == 1.0 and int(p["lead"].abs().sum()) == 0
    # the default (vectorised) order consumes the same generators differently: same ranges, another stream
    torch.manual_seed(43)
    np.random.seed(43)
    q = SyntheticFxBatcher(8, 88200, 44100, ("flanger",), torch.device("cpu")).sample_params()
    assert not np.array_equal(q["phase"].numpy(), p["phase"].numpy())
    for name, (lo, hi) in (("rate_hz", (0.5, 3.0)), ("feedback", (0.0, 0.7)), ("width", (0.25, 1.0))):
        assert float(q[name].min()) >= lo and float(q[name].max()) <= hi


def test_batcher_reference_order_phaser_items():
    """phaser items: rate, depth, centre, feedback, mix, crop offset per item (datasets.py:429-465,444)"""
    from mod_extraction_amd import util
    from mod_extraction_amd.data_modules import SyntheticFxBatcher, PHASER_FX
    torch.manual_seed(5)
    np.random.seed(5)
    b = SyntheticFxBatcher(4, 88200, 44100, ("phaser",), torch.device("cpu"), rng_order="reference")
    p = b.sample_params()
    torch.manual_seed(5)
    np.random.seed(5)
    for i in range(4):
        rate = util.sample_log_uniform(*PHASER_FX["rate_hz"])
        rate_n = int((44100 / rate) + 0.5)
        depth = util.sample_uniform(*PHASER_FX["depth"])
        centre = util.sample_log_uniform(*PHASER_FX["centre_frequency_hz"])
        fb = util.sample_uniform(*PHASER_FX["feedback"])
        mix = util.sample_uniform(*PHASER_FX["mix"])
        start = util.randint(0, rate_n + 1)
        got = [float(p[k][i]) for k in ("rate_hz", "depth", "centre_frequency_hz", "feedback", "mix")]
        want = [float(np.float32(v)) for v in (rate, depth, centre, fb, mix)]
        assert got == want and int(p["lead"][i]) == start and int(p["proc_extra"][i]) == rate_n
        assert p["shape"][i] == "cos" and abs(float(p["phase"][i]) - math.pi / 2) < 1e-6


def test_oracle_rand_mod_signal_and_tremolo_bit_exact(golden_dir):
    from oracle import fx as ofx, modulations as omod
    g = np.load(os.path.join(golden_dir, "rand_lfo_tremolo.npz"))
    shapes_gt = [SHAPES[i] for i in g["gt_shape"]]
    torch.manual_seed(7); np.random.seed(7)
    assert np.array_equal(omod.make_rand_mod_signal(6, 345, 172.5, 0.5, 3.0).numpy(), g["rand_a"])
    torch.manual_seed(8); np.random.seed(8)
    y = omod.make_rand_mod_signal(6, 345, 172.5, 0.5, 3.0, shapes_gt, None, torch.from_numpy(g["gt_phase"].copy()), 0.5,
                                  torch.from_numpy(g["gt_freq"].copy()), 0.25)
    assert np.array_equal(y.numpy(), g["rand_b"])
    y = omod.make_rand_mod_signal(6, 345, 172.5, 0.5, 3.0, shapes_gt, None, torch.from_numpy(g["gt_phase"].copy()), 0.0,
                                  torch.from_numpy(g["gt_freq"].copy()), 0.0)
    assert np.array_equal(y.numpy(), g["rand_c"])
    torch.manual_seed(9); np.random.seed(9)
    y = omod.make_rand_mod_signal(6, 345, 172.5, 0.5, 3.0, None, ["tri", "saw"], None, 0.5,
                                  torch.from_numpy(g["gt_freq"].copy()), 0.1)
    assert np.array_equal(y.numpy(), g["rand_d"])
    x, mod = torch.from_numpy(g["trem_x"]), torch.from_numpy(g["trem_mod"])
    assert np.array_equal(ofx.apply_tremolo(x, mod, 0.7).numpy(), g["trem_y_07"])
    assert np.array_equal(ofx.apply_tremolo(x, mod.unsqueeze(1).expand(-1, 2, -1), 1.0).numpy(), g["trem_y_10"])
    assert np.array_equal(ofx.apply_tremolo(x, mod, 0.0).numpy(), g["trem_y_00"])


def test_product_tremolo_on_the_host_matches_the_reference(golden_dir):
    """apply_tremolo is one torch expression: the same code path on CPU tensors is bit-identical to
    fx.py:13-22; the -m gpu twin runs it on the device."""
    from mod_extraction_amd import fx
    g = np.load(os.path.join(golden_dir, "rand_lfo_tremolo.npz"))
    x, mod = torch.from_numpy(g["trem_x"]), torch.from_numpy(g["trem_mod"])
    assert np.array_equal(fx.apply_tremolo(x, mod, 0.7).numpy(), g["trem_y_07"])
    assert np.array_equal(fx.apply_tremolo(x, mod.unsqueeze(1).expand(-1, 2, -1), 1.0).numpy(), g["trem_y_10"])
    with pytest.raises(AssertionError):
        fx.apply_tremolo(x, mod, 1.5)
    with pytest.raises(AssertionError):
        fx.apply_tremolo(x, mod[:2], 0.5)


def test_corner_helpers_of_one_row():
    """check_mod_sig / corners_to_mod_sig (modulations.py:241-257,311-343): host bookkeeping on one LFO row, checked on
    hand-built corner maps (the batched device forms are covered by the bit-exact corner tests on the GPU)."""
    from mod_extraction_amd import modulations as am
    n = 40
    top, bot = torch.zeros(n), torch.zeros(n)
    top[[10, 30]] = 1
    bot[[0, 20]] = 1
    m = am.corners_to_mod_sig(top, bot)
    assert m.shape == (n,) and float(m[10]) == 1.0 and float(m[20]) == 0.0 and float(m[30]) == 1.0
    assert torch.allclose(m[:11], torch.linspace(0, 1, 11)) and torch.allclose(m[20:31], torch.linspace(0, 1, 11))
    assert float(m[31:].abs().max()) == 0.0                       # nothing after the last corner
    assert float(am.corners_to_mod_sig(top, torch.zeros(n)).abs().max()) == 0.0
    sig = torch.rand(n)
    assert am.check_mod_sig(sig, top, bot) is True                # 2 + 2 corners, 20 frames apart (>= int(0.1 * 40) = 4)
    close = torch.zeros(n); close[[10, 12]] = 1
    assert am.check_mod_sig(sig, close, bot) is False             # two top corners 2 frames apart
    assert am.check_mod_sig(sig, torch.zeros(n), bot) is False    # no top corner
    many = torch.zeros(n); many[::5] = 1
    assert am.check_mod_sig(sig, many, bot) is False              # 8 top corners > 6
