"""Block-2 forward convolution alone (B clips, H = 128, T = 1) on the shipped library, for PMC / wall A/Bs of the MFMA shape:
    MODEX_MFMA_SHAPE=32 python tools/exp_shape_ab.py [B] [launches]      (v_mfma_f32_32x32x16_f16)
    MODEX_MFMA_SHAPE=16 python tools/exp_shape_ab.py [B] [launches]      (v_mfma_f32_16x16x32_f16, the default)
Random operands (the chip's clock under an MFMA stream depends on the data)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mod_extraction_amd import _hip  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
H, T = 128, 1
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
x = torch.randn((B, H, 4, 352, 16), generator=g).to(dev)
x_hi = x.half()
x_lo = (x - x_hi.float()).half()
w = (torch.randn((4 * 5 * 13 * 64 * 16,), generator=g) * 8).to(dev)
w_hi = w.half()
w_lo = (w - w_hi.float()).half()
bias = torch.zeros(64, device=dev)
out = torch.empty((B, 64, H // 2, 352), device=dev)
am = torch.empty((B, 64, H // 2, 352), device=dev, dtype=torch.uint8)
st = _hip.stream()


def fwd():
    _hip.call("mx_conv_block_fwd_f16", _hip.ptr(x_hi), _hip.ptr(x_lo), _hip.ptr(w_hi), _hip.ptr(w_lo), _hip.ptr(bias), B, H, 345, T,
              _hip.ptr(out), _hip.ptr(am), None, None, st)


for _ in range(3):
    fwd()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(N):
    fwd()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / N
print(f"shape {os.environ.get('MODEX_MFMA_SHAPE', '16')}: B={B} {ms:.3f} ms per launch, {2.0 * 64 * 64 * 65 * B * H * 345 / ms / 1e9:.1f} TFLOP/s algorithmic, chk {float(out.flatten()[12345]):.5g}")
