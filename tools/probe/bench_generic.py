"""GPU box: throughput of the GENERAL Spectral2DCNN path (class defaults of models.py:129-145: pool (3,1), five 64-channel blocks,
temp dilations 1..16, in_ch 1, 256 mels, 2 s clips) forward + backward, next to the shipped family on its fused kernels."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from mod_extraction_amd import models as am
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
def run(net, x, n=3):
    net = net.to(dev).train()
    def step():
        net.zero_grad()
        o, l = net(x)
        (o.sum() + 0.1 * l.sum()).backward()
    step(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(n): step()
    torch.cuda.synchronize()
    return (time.time() - t) / n
x1 = torch.rand(B, 1, 88200, device=dev) * 2 - 1
x2 = torch.rand(B, 2, 88200, device=dev) * 2 - 1
t_gen = run(am.Spectral2DCNN(), x1)
t_fam = run(am.Spectral2DCNN(in_ch=2, out_channels=[64] * 6, temp_dilations=[1, 1, 2, 4, 8, 16], pool_size=(2, 1)), x2)
print(f"B = {B} x 2 s: general path (class defaults) {t_gen * 1e3:.1f} ms per forward + backward = {B * 2 / t_gen:.0f} audio-s/s; "
      f"shipped family (fused f16x3 kernels) {t_fam * 1e3:.1f} ms = {B * 2 / t_fam:.0f} audio-s/s")
