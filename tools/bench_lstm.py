"""GPU-box timing of the LSTM-64 kernels alone (K10): forward with / without the BPTT stash, the serial
backward + weight-gradient GEMM, at the config-4 geometry (128 clips x 1024-sample chunks).
    python tools/bench_lstm.py [B] [T]
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mod_extraction_amd import _hip, models

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda:0")
torch.manual_seed(0)
em = models.LSTMEffectModel().to(dev)
x = torch.rand(B, 1, T, device=dev) * 2 - 1
lat = torch.rand(B, 1, T, device=dev)
wet = torch.rand(B, 1, T, device=dev) * 2 - 1
stash = torch.empty(B, T, 384, device=dev)
grad = torch.zeros(models.LSTM_NPARAM, device=dev)


def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


em.clear_hidden()
out = {"B": B, "T": T}
out["fwd_nostash_ms"] = timeit(lambda: em.run_chunk(x, lat))
out["fwd_stash_ms"] = timeit(lambda: em.run_chunk(x, lat, stash))
y, h0, c0 = em.run_chunk(x, lat, stash)
out["bwd_total_ms"] = timeit(lambda: em.bptt_l1_chunk(x, lat, y, wet, stash, h0, c0, 1.0 / (B * T), grad))
out["fwd_ns_per_step"] = out["fwd_stash_ms"] * 1e6 / T
out["bwd_ns_per_step"] = out["bwd_total_ms"] * 1e6 / T
print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in out.items()}))
