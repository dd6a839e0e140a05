"""Worker of tests/test_gpu_step.py::test_two_rank_step_equals_the_single_process_step_on_the_joined_batch.
Launched by torch.distributed.run with WORLD_SIZE ranks sharing one GPU (MODEX_SHARE_GPU=1, gloo): every rank takes its
slice of a fixed batch, runs ONE trainer.train_step (forward, backward, flat-gradient all-reduce, AdamW with 1/world) and
rank 0 saves the parameters; with WORLD_SIZE=1 the same script runs the whole batch in one process."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mod_extraction_amd import lightning, models, optim, trainer  # noqa: E402

out_path, total = sys.argv[1], int(sys.argv[2])
mode = sys.argv[3] if len(sys.argv) > 3 else "lfo"
env = trainer.init_distributed()
rank, world = env["rank"], env["world_size"]
dev = torch.device("cuda", env["local_rank"])
torch.cuda.set_device(dev)
if mode == "tbptt":
    # effect modelling: LSTM-64 under truncated BPTT with the ground-truth LFO (no extractor), 4 optimizer steps per batch
    n = 5200
    torch.manual_seed(4321); np.random.seed(4321)
    em = models.LSTMEffectModel()
    module = lightning.TBPTTLFOEffectModeling(1024, 1024, em, lfo_model=None, model_smooth_n_frames=0, should_stretch=False,
                                              discard_invalid_lfos=False, loss_dict={"l1": 1.0, "esr": 0.0, "dc": 0.0}).to(dev).train()
    opt = optim.FlatAdamW(module.parameters(), lr=1e-3, betas=(0.8, 0.99))
    g = torch.Generator().manual_seed(77)
    dry = torch.rand(total, 1, n, generator=g) * 1.6 - 0.8
    wet = (0.7 * dry + 0.2 * torch.roll(dry, 5, -1)).clamp(-1, 1)
    t = torch.arange(64) / 64.0
    mod = torch.stack([0.5 + 0.5 * torch.cos(2 * np.pi * (1.0 + 0.5 * i) * t + 0.4 * i) for i in range(total)])
    per = total // world
    sl = slice(rank * per, (rank + 1) * per)
    loss = module.training_step((dry[sl].to(dev), wet[sl].to(dev), mod[sl].to(dev), None), 0, optimizer=opt, world_size=world)
    torch.cuda.synchronize()
    if rank == 0:
        torch.save({"param": opt.flat_param.cpu(), "grad": opt.flat_grad.cpu() / world, "loss": float(loss), "world": world,
                    "steps": opt.step_count}, out_path)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    sys.exit(0)
n, sr = 22272, 44100
cfg = dict(in_ch=2, n_samples=n, sr=sr, n_fft=1024, hop_len=256, n_mels=64, kernel_size=(5, 13), out_channels=[64] * 6,
           temp_dilations=[1, 1, 2, 4, 8, 16], pool_size=(2, 1), latent_dim=1, use_ln=True)
torch.manual_seed(1234); np.random.seed(1234)                     # same weights and same full batch on every rank
module = lightning.LFOExtraction(models.Spectral2DCNN(**cfg), sr=sr, use_dry=True, model_smooth_n_frames=0,
                                 loss_dict={"l1": 1.0, "fdl1": 5.0, "sdl1": 10.0, "mse": 0.0}).to(dev).eval()
opt = optim.FlatAdamW(module.parameters(), lr=1e-3, betas=(0.8, 0.99))
g = torch.Generator().manual_seed(99)
dry = torch.rand(total, 1, n, generator=g) * 1.6 - 0.8
wet = (0.6 * dry + 0.3 * torch.roll(dry, 9, -1)).clamp(-1, 1)
t = torch.arange(882) / 441.0
mod = torch.stack([0.5 + 0.5 * torch.cos(2 * np.pi * (0.7 + 0.4 * i) * t + 0.3 * i) for i in range(total)])
per = total // world
sl = slice(rank * per, (rank + 1) * per)
batch = (dry[sl].to(dev), wet[sl].to(dev), mod[sl].to(dev), None)
loss = trainer.Trainer(log_fn=None).train_step(module, opt, batch)
torch.cuda.synchronize()
if rank == 0:
    torch.save({"param": opt.flat_param.cpu(), "grad": opt.flat_grad.cpu() / world, "loss": float(loss.detach()), "world": world},
               out_path)
if world > 1:
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
