"""GPU-box diagnostic: per-parameter and per-block gradient errors of the HIP CNN stack vs the CPU oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mod_extraction_amd import models as am
from tests.test_gpu_cnn import make_pair, audio, rel_err, _loss
n_samples, n_mels, B = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (22272, 64, 3)
dev = torch.device("cuda:0")
ref, mine = make_pair(dev, n_samples=n_samples, n_mels=n_mels)
ref.eval(); mine.eval()
x = audio(B, n_samples); masks = (3, 11, 20, 41)
# oracle with taps on every pool output and every LayerNorm output
taps = {}
h = ref.log_mel(x, masks)
for i, m in enumerate(ref.cnn):
    h = m(h)
    if isinstance(m, (torch.nn.MaxPool2d, torch.nn.LayerNorm)):
        taps[i] = h
        if h.requires_grad: h.retain_grad()
lat_r = h.mean(dim=-2); out_r = torch.sigmoid(ref.output(lat_r))
(_loss(out_r) + 0.1 * _loss(lat_r)).backward()
am.DEBUG_TAP = {}
out_m, lat_m = mine(x.to(dev), masks); (_loss(out_m) + 0.1 * _loss(lat_m)).backward()
W = mine.n_frames
for l in range(5, -1, -1):
    g_r = taps[4 * l + 2].grad
    print(f"block {l}: G rel err {rel_err(am.DEBUG_TAP[f'G{l}'].cpu()[..., :W], g_r):.3e}", end="")
    if l > 0:
        print(f"   dxhat rel err {rel_err(am.DEBUG_TAP[f'dxhat{l}'].cpu()[..., :W], taps[4 * l].grad):.3e}")
    else:
        print()
gr = dict(ref.named_parameters())
for name, p in mine.named_parameters():
    g, r = p.grad.cpu(), gr[name].grad
    print(f"{name:16s} rel err {rel_err(g, r):.3e}  max|ref| {float(r.abs().max()):.3e}")
# --- isolate block 2: recompute its dgrad with my saved amax and with the oracle's argmax
from mod_extraction_amd import _hip
l = 2
z = taps[4 * l + 1] if (4 * l + 1) in taps else None
hh = ref.log_mel(x, masks)
for i, m in enumerate(ref.cnn):
    hin = hh
    hh = m(hh)
    if i == 4 * l + 1:
        zz = hh.detach()
am_r = (zz[:, :, 1::2] > zz[:, :, 0::2]).to(torch.uint8)
am_m = am.DEBUG_TAP[f"amax{l}"].cpu()[..., :W]
print("block2 amax mismatches vs oracle:", int((am_m != am_r).sum()), "of", am_r.numel(), " values in mine:", am_m.unique().tolist())
mm = (am_m != am_r)
if mm.any():
    idx = mm.nonzero()
    print(" first mismatches (b,c,h,w):", idx[:8].tolist())
    print(" mismatch count by w:", mm.sum(dim=(0, 1, 2)).tolist())
