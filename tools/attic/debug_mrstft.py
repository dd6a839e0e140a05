import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import losses as olosses
from mod_extraction_amd import losses as alosses
dev = torch.device("cuda:0")
for B, T in ((3, 8000), (2, 88200)):
    torch.manual_seed(B + T)
    t = torch.arange(T) / 44100.0
    y = (0.5 * torch.sin(2 * np.pi * 330.0 * t) + 0.2 * torch.rand(B, 1, T) - 0.1).clamp(-1, 1)
    x0 = (0.8 * y + 0.1 * torch.roll(y, 7, -1) + 0.05 * torch.randn(B, 1, T)).clamp(-1, 1)
    class MR64(olosses.MultiResolutionSTFTLoss):
        def _mag(self, x, n_fft, hop, win):
            s = torch.stft(x.reshape(-1, x.size(-1)), n_fft, hop, win, torch.hann_window(win, dtype=torch.float64), return_complex=True)
            return torch.sqrt(torch.clamp(s.real ** 2 + s.imag ** 2, min=self.eps))
    x64 = x0.double().requires_grad_(True); l64 = MR64()(x64, y.double()); l64.backward()
    x32 = x0.clone().requires_grad_(True); l32 = olosses.MultiResolutionSTFTLoss()(x32, y); l32.backward()
    xd = x0.to(dev).requires_grad_(True); lm = alosses.get_loss_func_by_name("mrstft")(xd, y.to(dev)); lm.backward()
    g64 = x64.grad; mx = g64.abs().max()
    print(B, T, "loss64", float(l64), "loss32", float(l32), "mine", float(lm))
    print("  grad err vs fp64: oracle32", float((x32.grad.double() - g64).abs().max() / mx), " mine", float((xd.grad.cpu().double() - g64).abs().max() / mx),
          " rms: oracle32", float((x32.grad.double() - g64).pow(2).mean().sqrt() / g64.pow(2).mean().sqrt()), " mine", float((xd.grad.cpu().double() - g64).pow(2).mean().sqrt() / g64.pow(2).mean().sqrt()))
