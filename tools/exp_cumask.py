"""Experiment: does confining the prefetch work (batch render + frozen extractor, side stream) and the TBPTT recurrence
(main stream) to disjoint CU sets reduce their interference in config 4?  Streams with CU masks come from
hipExtStreamCreateWithCUMask (ctypes) and are wrapped as torch external streams.
    python tools/exp_cumask.py            # prints ms per batch for several mask layouts
"""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from mod_extraction_amd import data_modules, lightning, models, optim

hip = ctypes.CDLL("libamdhip64.so")
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
torch.zeros(1, device=dev)


def masked_stream(words):
    st = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), len(words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value, device=dev)


CNN_CFG = dict(in_ch=2, n_samples=88200, sr=44100, n_fft=1024, hop_len=256, n_mels=256, kernel_size=(5, 13),
               out_channels=[64] * 6, temp_dilations=[1, 1, 2, 4, 8, 16], pool_size=(2, 1), latent_dim=1)


def run(main_words, side_words, steps=4):
    torch.manual_seed(44); np.random.seed(44)
    cnn = models.Spectral2DCNN(**CNN_CFG)
    em = models.LSTMEffectModel()
    mod = lightning.TBPTTLFOEffectModeling(1024, 1024, em, lfo_model=cnn, discard_invalid_lfos=False,
                                           loss_dict={"l1": 1.0, "esr": 0.0, "dc": 0.0}).to(dev).train()
    opt = optim.FlatAdamW([p for p in mod.parameters() if p.requires_grad], lr=1e-4, betas=(0.8, 0.99))
    bt = data_modules.SyntheticFxBatcher(128, 88200, 44100, ("phaser",), dev, audio_seed=44, overlap=True)
    if side_words is not None:
        bt._side = masked_stream(side_words)
    bt.ahead_fn = lambda b: mod.prepare_ahead((b[0], b[1], None, None))
    main = masked_stream(main_words) if main_words is not None else torch.cuda.current_stream()

    def step():
        dry, wet, _, _ = bt.next_batch()
        mod.training_step((dry, wet, None, None), 0, optimizer=opt, world_size=1, prep=bt.last_ahead)

    with torch.cuda.stream(main):
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


FULL = [0xFFFFFFFF] * 8
F = 0xFFFFFFFF
layouts = {
    "no masks": (None, None),
    "main words 0-3 / side words 4-7": ([F] * 4 + [0] * 4, [0] * 4 + [F] * 4),
    "main words 0-4 / side words 5-7": ([F] * 5 + [0] * 3, [0] * 5 + [F] * 3),
    "main words 0-5 / side words 6-7": ([F] * 6 + [0] * 2, [0] * 6 + [F] * 2),
    "main words 0-3 / side words 4-6": ([F] * 4 + [0] * 4, [0] * 4 + [F] * 3 + [0]),
    "main words 0-3 / side unmasked": ([F] * 4 + [0] * 4, None),
}
for name, (m, s) in layouts.items():
    try:
        print(f"{name:50s} {run(m, s):7.2f} ms per batch", flush=True)
    except Exception as e:
        print(name, "FAILED", repr(e)[:200], flush=True)
