"""GPU-box diagnostic: routed-pool oracle comparison, all parameters, fp64 oracle as arbiter."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, copy
from mod_extraction_amd import models as am
from tests.test_gpu_cnn import make_pair, audio, rel_err, _loss, oracle_forward_routed
n_samples, n_mels, B, in_ch = (int(v) for v in sys.argv[1:5])
precision = sys.argv[5] if len(sys.argv) > 5 else 'f16x3'
dev = torch.device("cuda:0")
ref, mine = make_pair(dev, n_samples=n_samples, n_mels=n_mels, in_ch=in_ch)
ref.eval(); mine.eval(); mine.conv_precision = precision
x = audio(B, n_samples); masks = (3, 11, 20, 41)
if in_ch == 1: x = x[:, 1:2]
am.DEBUG_TAP = {}
out_m, lat_m = mine(x.to(dev), masks); (_loss(out_m) + 0.1 * _loss(lat_m)).backward()
amax = [am.DEBUG_TAP[f"amax{l}"] for l in range(6)]
out_r, lat_r, n_ties = oracle_forward_routed(ref, x, masks, am.DEBUG_TAP, mine.n_frames)
(_loss(out_r) + 0.1 * _loss(lat_r)).backward()
ref64 = copy.deepcopy(ref).double()
for p in ref64.parameters(): p.grad = None
lm = ref.log_mel(x, masks).double()          # same fp32 log-mel, everything after in fp64
h = lm
for i, m in enumerate(ref64.cnn):
    if isinstance(m, torch.nn.MaxPool2d):
        pick = amax[i // 4].cpu()[..., :mine.n_frames].bool(); h = torch.where(pick, h[:, :, 1::2], h[:, :, 0::2])
    elif isinstance(m, torch.nn.PReLU):
        pos = am.DEBUG_TAP[f'p{i // 4}'].cpu()[..., :mine.n_frames] > 0; h = torch.where(pos, h, m.weight.view(1, -1, 1, 1) * h)
    else:
        h = m(h)
lat64 = h.mean(dim=-2); out64 = torch.sigmoid(ref64.output(lat64))
(_loss(out64) + 0.1 * _loss(lat64)).backward()
print("ties", n_ties, "out err fp32-oracle", rel_err(out_m.detach().cpu(), out_r.detach()),
      " vs fp64:", rel_err(out_m.detach().cpu().double(), out64.detach()), " oracle32 vs fp64:", rel_err(out_r.detach().double(), out64.detach()))
g32 = dict(ref.named_parameters()); g64 = dict(ref64.named_parameters())
for name, p in mine.named_parameters():
    g = p.grad.cpu()
    print(f"{name:16s} mine-vs-oracle32 {rel_err(g, g32[name].grad):.2e}   mine-vs-fp64 {rel_err(g.double(), g64[name].grad):.2e}"
          f"   oracle32-vs-fp64 {rel_err(g32[name].grad.double(), g64[name].grad):.2e}")
