"""Experiment: the TBPTT loop of config 4 alone (83 optimizer steps on a prepared batch, nothing on the side stream),
against the sum of its kernels timed alone (tools/bench_lstm.py) -- how much of a batch is launch boundaries?
    python tools/exp_tbptt_only.py
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from mod_extraction_amd import data_modules, lightning, models, optim, streams

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
CNN_CFG = dict(in_ch=2, n_samples=88200, sr=44100, n_fft=1024, hop_len=256, n_mels=256, kernel_size=(5, 13),
               out_channels=[64] * 6, temp_dilations=[1, 1, 2, 4, 8, 16], pool_size=(2, 1), latent_dim=1)
torch.manual_seed(44); np.random.seed(44)
cnn = models.Spectral2DCNN(**CNN_CFG)
em = models.LSTMEffectModel()
mod = lightning.TBPTTLFOEffectModeling(1024, 1024, em, lfo_model=cnn, discard_invalid_lfos=False,
                                       loss_dict={"l1": 1.0, "esr": 0.0, "dc": 0.0}).to(dev).train()
opt = optim.FlatAdamW([p for p in mod.parameters() if p.requires_grad], lr=1e-4, betas=(0.8, 0.99))
bt = data_modules.SyntheticFxBatcher(128, 88200, 44100, ("phaser",), dev, audio_seed=44, overlap=False)
dry, wet, _, _ = bt.next_batch()
batch = (dry, wet, None, None)
prep = mod.prepare_ahead(batch)
torch.cuda.synchronize()


def loop(n=4):
    mod.training_step(batch, 0, optimizer=opt, world_size=1, prep=prep)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        mod.training_step(batch, 0, optimizer=opt, world_size=1, prep=prep)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print(f"unmasked stream            {loop():7.2f} ms per batch of 83 optimizer steps")
main, side = streams.cu_partition(dev)
with torch.cuda.stream(main):
    print(f"main stream of the CU partition (160 CUs) {loop():7.2f} ms")
# CPU time of the same loop when the GPU work is enqueued without waiting
t0 = time.perf_counter()
mod.training_step(batch, 0, optimizer=opt, world_size=1, prep=prep)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"host time to enqueue one batch {1e3 * (t1 - t0):7.2f} ms")
