"""Diagnostic for tests/test_gpu_step.py::test_lfo_extraction_twenty_step_trajectory_vs_oracle: per step the loss of both
trajectories, the per-tensor parameter distance (in units of lr) and -- with the oracle's parameters copied onto the device
before the step (teacher forcing, TF=1) -- the per-tensor gradient error of that step."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import lightning as ol, models as om
from mod_extraction_amd import data_modules, lightning, models, optim, trainer

TF = int(os.environ.get("TF", "0"))
MASK = float(os.environ.get("MASK", "0.25"))
dev = torch.device("cuda:0")
n, sr, B, steps, lr = 22272, 44100, 4, int(os.environ.get("STEPS", "8")), 1e-4
cfg = dict(in_ch=2, n_samples=n, sr=sr, n_fft=1024, hop_len=256, n_mels=64, kernel_size=(5, 13), out_channels=[64] * 6,
           temp_dilations=[1, 1, 2, 4, 8, 16], pool_size=(2, 1), latent_dim=1, freq_mask_amount=MASK, time_mask_amount=MASK, use_ln=True)
loss_dict = {"l1": 1.0, "fdl1": 5.0, "sdl1": 10.0, "mse": 0.0}
torch.manual_seed(11); np.random.seed(11)
ref = om.Spectral2DCNN(**cfg).train()
mine = models.Spectral2DCNN(**cfg)
mine.load_state_dict(ref.state_dict())
module = lightning.LFOExtraction(mine, sr=sr, model_smooth_n_frames=0, loss_dict=loss_dict).to(dev).train()
opt = optim.FlatAdamW(module.parameters(), lr=lr, betas=(0.8, 0.99))
ref_opt = torch.optim.AdamW(ref.parameters(), lr=lr, betas=(0.8, 0.99))
batcher = data_modules.SyntheticFxBatcher(B, n, sr, ("flanger", "chorus", "phaser"), dev, audio_seed=7)
runner = trainer.Trainer(log_fn=None)
names = [k for k, _ in ref.named_parameters()]
for i in range(steps):
    dry, wet, mod, _ = batcher.render(batcher.sample_params())
    if TF:
        with torch.no_grad():
            for (_, p), (_, q) in zip(mine.named_parameters(), ref.named_parameters()):
                p.copy_(q.to(dev))
    torch.manual_seed(1000 + i)
    loss_r, _ = ol.lfo_train_step(ref, ref_opt, dry.cpu(), wet.cpu(), mod.cpu(), loss_dict)
    torch.manual_seed(1000 + i)
    loss = float(runner.train_step(module, opt, (dry, wet, mod, None)))
    gerr = []
    for (k, p), (_, q) in zip(mine.named_parameters(), ref.named_parameters()):
        g, h = p.grad.cpu().double(), q.grad.double()
        gerr.append((float((g - h).abs().max() / h.abs().max().clamp_min(1e-30)), k))
    pd = []
    for (k, p), (_, q) in zip(mine.named_parameters(), ref.named_parameters()):
        pd.append((float((p.detach().cpu() - q.detach()).abs().max()) / lr, k))
    print(f"step {i}: loss hip {loss:.7f} oracle {loss_r:.7f} rel {abs(loss - loss_r) / abs(loss_r):.2e} | worst grad rel err "
          f"{max(gerr)[0]:.2e} ({max(gerr)[1]}) | worst param diff {max(pd)[0]:.3f} lr ({max(pd)[1]})")
    if os.environ.get("VERBOSE"):
        print("   grad:", " ".join(f"{k}:{e:.1e}" for e, k in gerr))
        print("   par :", " ".join(f"{k}:{e:.2f}" for e, k in pd))
