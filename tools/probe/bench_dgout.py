import json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from mod_extraction_amd import _hip, models
B, T = 128, 1024
dev = torch.device("cuda:0")
torch.manual_seed(0)
em = models.LSTMEffectModel().to(dev)
x = torch.rand(B, 1, T, device=dev) * 2 - 1
lat = torch.rand(B, 1, T, device=dev)
wet = torch.rand(B, 1, T, device=dev) * 2 - 1
stash = torch.empty(B, T, 384, device=dev)
grad = torch.zeros(models.LSTM_NPARAM, device=dev)
def timeit(fn, n=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
em.clear_hidden()
y, h0, c0 = em.run_chunk(x, lat, stash)
for rep in range(3):
    t1 = timeit(lambda: em.bptt_l1_chunk(x, lat, y, wet, stash, h0, c0, 1.0 / (B * T), grad))
    t2 = timeit(lambda: em.bptt_chunk_dlfo(x, lat, y, stash, h0, c0, grad, wet=wet, loss_scale=1.0 / (B * T)))
    print(f"bwd_l1 {t1*1e3:.1f} us   bwd_dgate+dlfo {t2*1e3:.1f} us")
from mod_extraction_amd.models import _rows, LSTM_NPARAM
part = torch.empty((B, LSTM_NPARAM), device=dev)
dgate = torch.empty((B, T, 256), device=dev)
xp, xs = _rows(x); lp, ls = _rows(lat); yp, ys = _rows(y); wp, ws = _rows(wet)
whh, fcw = em.lstm.weight_hh_l0.detach().contiguous(), em.fc.weight.detach().contiguous()
def k_dg():
    _hip.call("mx_lstm_bwd_dgate", xp, xs, lp, ls, yp, ys, wp, ws, None, 0, _hip.ptr(stash), _hip.ptr(whh), _hip.ptr(fcw), _hip.ptr(h0), _hip.ptr(c0),
              1.0 / (B * T), _hip.ptr(part), _hip.ptr(dgate), B, T, _hip.stream())
def k_l1():
    _hip.call("mx_lstm_bwd_l1", xp, xs, lp, ls, yp, ys, wp, ws, _hip.ptr(stash), _hip.ptr(whh), _hip.ptr(fcw), _hip.ptr(h0), _hip.ptr(c0),
              1.0 / (B * T), _hip.ptr(part), B, T, _hip.stream())
for rep in range(3):
    print(f"kernel only: bwd_l1 {timeit(k_l1)*1e3:.1f} us   bwd_dgate {timeit(k_dg)*1e3:.1f} us")
