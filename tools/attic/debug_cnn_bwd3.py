"""GPU-box diagnostic: per-block G / dxhat errors against the routed fp64 oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, copy
from mod_extraction_amd import models as am
from tests.test_gpu_cnn import make_pair, audio, rel_err, _loss
n_samples, n_mels, B, in_ch = (int(v) for v in sys.argv[1:5])
dev = torch.device("cuda:0")
ref, mine = make_pair(dev, n_samples=n_samples, n_mels=n_mels, in_ch=in_ch)
ref.eval(); mine.eval()
x = audio(B, n_samples); masks = (3, 11, 20, 41)
if in_ch == 1: x = x[:, 1:2]
am.DEBUG_TAP = {}
out_m, lat_m = mine(x.to(dev), masks); (_loss(out_m) + 0.1 * _loss(lat_m)).backward()
tap = am.DEBUG_TAP
W = mine.n_frames
ref64 = copy.deepcopy(ref).double()
h = ref.log_mel(x, masks).double()
pools, lns = {}, {}
for i, m in enumerate(ref64.cnn):
    if isinstance(m, torch.nn.MaxPool2d):
        pick = tap[f"amax{i // 4}"].cpu()[..., :W].bool(); h = torch.where(pick, h[:, :, 1::2], h[:, :, 0::2])
        h.retain_grad(); pools[i // 4] = h
    else:
        h = m(h)
        if isinstance(m, torch.nn.LayerNorm) and h.requires_grad:
            h.retain_grad(); lns[i // 4] = h
lat64 = h.mean(dim=-2); out64 = torch.sigmoid(ref64.output(lat64))
(_loss(out64) + 0.1 * _loss(lat64)).backward()
for l in range(5, -1, -1):
    g = tap[f"G{l}"].cpu()[..., :W].double(); r = pools[l].grad
    e = (g - r).abs()
    print(f"block {l}: G rel err {rel_err(g, r):.2e}  sum(G) mine {float(g.sum()):+.6e} ref {float(r.sum()):+.6e}"
          f"  mean|err| {float(e.mean()):.2e} max|G| {float(r.abs().max()):.2e}")
    if l in lns:
        d = tap[f"dxhat{l}"].cpu()[..., :W].double(); rd = lns[l].grad
        ed = (d - rd).abs()
        print(f"         dxhat rel err {rel_err(d, rd):.2e}  err by column-quartile:",
              [f"{float(ed[..., q * W // 4:(q + 1) * W // 4].max()):.1e}" for q in range(4)],
              " err by row:", [f"{float(ed[:, :, r_].max()):.1e}" for r_ in range(min(8, ed.size(2)))])
