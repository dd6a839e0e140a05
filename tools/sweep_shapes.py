import sys, torch, numpy as np
sys.path.insert(0, ".")
from mod_extraction_amd import models
dev = torch.device("cuda:0")
def rel(a, b): return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
for (B, n, mels, in_ch) in [(1, 44100, 128, 2), (5, 30000, 64, 2), (7, 88200, 256, 2), (2, 88200, 256, 1), (9, 20000, 64, 2), (16, 66150, 128, 2)]:
    cfg = dict(in_ch=in_ch, n_samples=n, sr=44100, n_fft=1024, hop_len=256, n_mels=mels, kernel_size=(5, 13),
               out_channels=[64] * 6, temp_dilations=[1, 1, 2, 4, 8, 16], pool_size=(2, 1), latent_dim=1,
               freq_mask_amount=0.25, time_mask_amount=0.25, use_ln=True)
    torch.manual_seed(0)
    m = models.Spectral2DCNN(**cfg).to(dev).eval()
    g = torch.Generator().manual_seed(1)
    x = (torch.rand(B, in_ch, n, generator=g) * 2 - 1).to(dev)
    res = {}
    for prec in ("f16x3", "f32"):
        m.conv_precision = prec
        m.zero_grad(set_to_none=True)
        out, lat = m(x)
        w = torch.linspace(0.5, 1.5, out.numel(), device=dev).view_as(out)
        ((out * w).sum() / out.numel() + 0.1 * lat.mean()).backward()
        res[prec] = (out.detach().clone(), lat.detach().clone(), [p.grad.detach().clone() for p in m.parameters()])
    eo, el = rel(res["f16x3"][0], res["f32"][0]), rel(res["f16x3"][1], res["f32"][1])
    eg = max(rel(a, b) for a, b in zip(res["f16x3"][2], res["f32"][2]))
    print(f"B={B} n={n} mels={mels} in_ch={in_ch}: out {eo:.2e} latent {el:.2e} grads {eg:.2e}", "OK" if eo < 1e-5 and el < 1e-5 else "CHECK", flush=True)
