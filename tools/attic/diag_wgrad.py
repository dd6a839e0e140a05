"""GPU-box diagnostic (not part of the product): builds wgrad.hip with -DWG_DIAG into tools/diag/libwgdiag.so,
runs the block-2 weight-gradient launch of the headline workload and prints where wave 0 of the
workgroups spends its cycles (barrier wait / LDS store / load issue + barrier / MFMA loop)."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
so = os.path.join(ROOT, "tools", "diag", "libwgdiag.so")
if not os.path.exists(so) or "--rebuild" in sys.argv:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                           "-DWG_DIAG", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "mod_extraction_amd/csrc/wgrad.hip"), "-o", so])
lib = ctypes.CDLL(so)
B, H, T = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 and sys.argv[1].isdigit() else (64, 128, 1)
dev = torch.device("cuda:0")
torch.manual_seed(0)
P = 352
G = torch.randn(B, 64, H // 2, P, device=dev); amax = torch.randint(0, 2, (B, 64, H // 2, P), device=dev, dtype=torch.uint8)
x = torch.randn(B, 64, H, P, device=dev); stats = torch.rand(B, 64, 2, device=dev) + 0.5; slope = torch.rand(64, device=dev) * 0.3
rows = B * H; rps = max(1, -(-rows // 256)); n_slabs = -(-rows // rps)
part = torch.empty(n_slabs * 65 * 64 * 64, device=dev); dW = torch.empty(64, 64, 5, 13, device=dev)
diag = torch.zeros(n_slabs * 10 * 4, device=dev, dtype=torch.int64)
lib.mx_diag_set_buffer(ctypes.c_void_p(diag.data_ptr()))
P_, I64, I32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
lib.mx_conv_block_wgrad.argtypes = [P_, P_, P_, P_, P_, I64, I64, I64, I64, I32, I64, P_, P_, P_]
for it in range(2):
    diag.zero_()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    rc = lib.mx_conv_block_wgrad(G.data_ptr(), amax.data_ptr(), x.data_ptr(), stats.data_ptr(), slope.data_ptr(), B, 64, H, 345, T, rps,
                                 part.data_ptr(), dW.data_ptr(), None)
    b.record(); torch.cuda.synchronize()
    assert rc == 0
d = diag.view(-1, 4).double().cpu()
tot = d.sum(1)
print(f"B={B} H={H} T={T}: {a.elapsed_time(b):.3f} ms, {d.shape[0]} workgroups, rows/slab {rps}")
names = ["barrier-1 wait", "LDS store", "issue + barrier-2", "MFMA loop"]
for i, n in enumerate(names):
    print(f"  {n:18s} {100 * float((d[:, i] / tot).mean()):5.1f} %   mean {float(d[:, i].mean()) / 1e3:9.1f} kcycles per workgroup")
print(f"  total per workgroup {float(tot.mean()) / 1e3:.1f} kcycles (memtime ticks)")
