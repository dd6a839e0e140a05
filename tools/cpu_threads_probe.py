import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import models as om, lightning as ol
import bench
torch.manual_seed(0)
B = 8
model = om.Spectral2DCNN(**bench.CNN_CFG); model.train()
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, betas=(0.8, 0.99))
dry = torch.rand(B, 1, 88200) * 2 - 1; wet = torch.rand(B, 1, 88200) * 2 - 1; mod = torch.rand(B, 882)
for nt in (16, 32, 64, 128, 256):
    torch.set_num_threads(nt)
    ol.lfo_train_step(model, opt, dry, wet, mod, bench.LOSS)
    t0 = time.perf_counter(); ol.lfo_train_step(model, opt, dry, wet, mod, bench.LOSS); dt = time.perf_counter() - t0
    print(nt, "threads:", round(dt, 2), "s/step ->", round(B * 2 / dt, 2), "audio-s/s", flush=True)
