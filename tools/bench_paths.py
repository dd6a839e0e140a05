"""GPU-box measurements of the non-headline pieces of the path (DESIGN.md section 5 quotes these):
  * effect kernels alone (flanger / chorus / phaser / log-mel) at bs = 256 x 2 s: ms, GB/s vs the
    12 / 8 B-per-sample algorithmic traffic
  * BASELINE config 4: effect-model TBPTT train batch (frozen CNN + LSTM-64, bs = 128 x 2 s)
  * BASELINE config 5 (per-GPU share): flanger + MR-STFT loss fwd/bwd at bs = 256 x 4 s
    python tools/bench_paths.py
"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from mod_extraction_amd import _hip, data_modules, fx, lightning, losses, models, modulations, optim

dev = torch.device("cuda:0")
SR = 44100


def timeit(fn, n=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def effects():
    B, N = 256, 88200
    out = {}
    torch.manual_seed(0); np.random.seed(0)
    for kind in ("flanger", "chorus", "phaser"):
        bt = data_modules.SyntheticFxBatcher(B, N, SR, (kind,), dev, audio_seed=1)
        p = bt.sample_params()
        bt.render(p)
        names = {"flanger": ["mx_flanger_fwd"], "chorus": ["mx_flanger_fwd"], "phaser": ["mx_phaser_fwd"]}[kind]
        with _hip.KernelTimer(set(names)) as kt:
            for _ in range(3): bt.render(p)
        ms = np.mean(list(kt.results().values())[0])
        n_proc = N + (float(p["lead"].float().mean()) if kind == "phaser" else 0)
        bytes_alg = B * N * (8 if kind != "phaser" else 8)          # x in, y out (LFO resampled in-kernel from 882 points)
        out[kind] = {"ms": round(ms, 3), "GBps_algorithmic": round(bytes_alg / ms / 1e6, 1),
                     "audio_s_per_s": round(B * 2.0 / (ms * 1e-3)), "ns_per_sample_per_clip": round(ms * 1e6 / n_proc, 2)}
    m = models.Spectral2DCNN(in_ch=2, n_mels=256, out_channels=[64] * 6, temp_dilations=[1, 1, 2, 4, 8, 16], pool_size=(2, 1)).to(dev)
    x = torch.rand(B, 2, N, device=dev) * 2 - 1
    ms = timeit(lambda: m.log_mel(x, (0, 0, 0, 0)))
    out["logmel"] = {"ms": round(ms, 3), "GBps_algorithmic": round(B * 1.413e6 / ms / 1e6, 1)}
    return out


def config4(B=128):
    torch.manual_seed(44); np.random.seed(44)
    cnn = models.Spectral2DCNN(in_ch=2, n_mels=256, out_channels=[64] * 6, temp_dilations=[1, 1, 2, 4, 8, 16], pool_size=(2, 1))
    em = models.LSTMEffectModel()
    mod = lightning.TBPTTLFOEffectModeling(1024, 1024, em, lfo_model=cnn, discard_invalid_lfos=False,
                                           loss_dict={"l1": 1.0, "esr": 0.0, "dc": 0.0}).to(dev).train()
    opt = optim.FlatAdamW([p for p in mod.parameters() if p.requires_grad], lr=1e-4, betas=(0.8, 0.99))
    bt = data_modules.SyntheticFxBatcher(B, 88200, SR, ("phaser",), dev, audio_seed=2)
    def step():
        dry, wet, _, _ = bt.next_batch()
        mod.training_step((dry, wet, None, None), 0, optimizer=opt, world_size=1)
    for _ in range(1): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with _hip.KernelTimer({"mx_lstm_fwd", "mx_lstm_bwd_l1"}) as kt:
        n = 2
        for _ in range(n): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    r = kt.results()
    return {"batch": B, "ms_per_batch": round(dt * 1e3, 1), "audio_s_per_s": round(B * 2.0 / dt, 1), "optimizer_steps_per_batch": 83,
            "lstm_fwd_ms_per_chunk": round(float(np.mean(r["mx_lstm_fwd"])), 3), "lstm_bwd_ms_per_chunk": round(float(np.mean(r["mx_lstm_bwd_l1"])), 3),
            "lstm_fwd_us_per_step": round(float(np.mean(r["mx_lstm_fwd"])) * 1e3 / 1024, 3)}


def config5(B=256, N=176400):
    torch.manual_seed(5); np.random.seed(5)
    bt = data_modules.SyntheticFxBatcher(B, N, SR, ("flanger",), dev, audio_seed=3)
    loss_fn = losses.get_loss_func_by_name("mrstft")
    def step():
        dry, wet, _, _ = bt.next_batch()
        pred = (0.9 * wet + 0.1 * dry).requires_grad_(True)
        loss_fn(pred, wet).backward()
    ms = timeit(step, n=3, warm=1)
    with _hip.KernelTimer({"mx_flanger_fwd", "mx_mrstft_loss"}) as kt:
        step()
    r = kt.results()
    return {"batch_per_gpu": B, "n_samples": N, "ms_per_step": round(ms, 2), "audio_s_per_s": round(B * 4.0 / (ms * 1e-3)),
            "flanger_ms": round(float(np.mean(r["mx_flanger_fwd"])), 3), "mrstft_ms": round(float(np.mean(r["mx_mrstft_loss"])), 3),
            "mrstft_GBps_algorithmic": round(B * N * 12 / float(np.mean(r["mx_mrstft_loss"])) / 1e6, 1),
            "flanger_GBps_algorithmic": round(B * N * 8 / float(np.mean(r["mx_flanger_fwd"])) / 1e6, 1)}


if __name__ == "__main__":
    which = sys.argv[1:] or ["effects", "config4", "config5"]
    if "effects" in which:
        print(json.dumps({"effects_bs256": effects()}))
    if "config4" in which:
        print(json.dumps({"config4_tbptt": config4()}))
    if "config5" in which:
        print(json.dumps({"config5_flanger_mrstft": config5()}))
