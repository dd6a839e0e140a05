"""GPU-box diagnostic: per-block forward errors of the HIP CNN stack against the CPU oracle.
    python tools/debug_cnn.py [n_samples n_mels B]
"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import models as omodels
from mod_extraction_amd import models as am, _hip
from tests.test_gpu_cnn import make_pair, audio, rel_err

n_samples, n_mels, B = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (22272, 64, 2)
dev = torch.device("cuda:0")
ref, mine = make_pair(dev, n_samples=n_samples, n_mels=n_mels)
ref.eval(); mine.eval()
x = audio(B, n_samples)
masks = (3, 11, 20, 41)
with torch.no_grad():
    lm_r = ref.log_mel(x, masks)
    lm_m = mine.log_mel(x.to(dev), masks)
    W = mine.n_frames
    print("logmel max abs err", float((lm_m.cpu()[..., :W] - lm_r).abs().max()))
    # feed the ORACLE's logmel into my stack so block errors are isolated
    cur = torch.zeros_like(lm_m); cur[..., :W] = lm_r.to(dev)
    slope, cin, H = None, 2, n_mels
    h_r = lm_r
    ps = mine._stack_params()
    for l in range(6):
        w, b, a = ps[3*l], ps[3*l+1], ps[3*l+2]
        stats = torch.empty((B, cin, 2), device=dev)
        _hip.call("mx_plane_stats", _hip.ptr(cur), _hip.ptr(slope), B, cin, H, W, am.LN_EPS, _hip.ptr(stats), _hip.stream())
        wt = am._pack(w, 0)
        p = torch.empty((B, 64, H // 2, am.PITCH), device=dev); amax = torch.empty((B, 64, H // 2, am.PITCH), device=dev, dtype=torch.uint8)
        _hip.call("mx_conv_block_fwd", _hip.ptr(cur), _hip.ptr(stats), _hip.ptr(slope), _hip.ptr(wt), _hip.ptr(b.contiguous()),
                  B, cin, H, W, int(mine.temp_dilations[l]), 1 if l == 0 else 0, _hip.ptr(p), _hip.ptr(amax), _hip.stream())
        torch.cuda.synchronize()
        ln, conv, pool, prelu = ref.cnn[4*l], ref.cnn[4*l+1], ref.cnn[4*l+2], ref.cnn[4*l+3]
        xin = h_r
        mean_r = xin.mean(dim=(-2, -1)); var_r = xin.var(dim=(-2, -1), unbiased=False)
        print(f"block {l}: stats mean err {float((stats[..., 0].cpu() - mean_r).abs().max()):.3e} rstd rel err "
              f"{float(((stats[..., 1].cpu() - (var_r + 1e-5).rsqrt()).abs() / (var_r + 1e-5).rsqrt()).max()):.3e}")
        z = conv(ln(xin)); pz = pool(z)
        print(f"   pooled preact rel err {rel_err(p.cpu()[..., :W], pz):.3e}  (max |ref| {float(pz.abs().max()):.3f})"
              f"  pad cols zero: {bool((p[..., W:] == 0).all())}")
        am_r = (z[:, :, 1::2] > z[:, :, 0::2]).to(torch.uint8)
        print(f"   argmax mismatches {int((amax.cpu()[..., :W] != am_r).sum())} of {am_r.numel()}")
        h_r = prelu(pz)
        cur = torch.zeros_like(p); cur[..., :W] = pz.to(dev)    # again feed oracle values forward
        slope, cin, H = a.contiguous(), 64, H // 2
t = time.time(); out_m, lat_m = mine(x.to(dev), masks); torch.cuda.synchronize(); print("fwd time", time.time() - t)
out_r, lat_r = ref(x, masks)
print("end-to-end out rel err", rel_err(out_m.detach().cpu(), out_r.detach()), "latent", rel_err(lat_m.detach().cpu(), lat_r.detach()))
