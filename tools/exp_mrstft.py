"""Variant harness for csrc/mrstft.hip: build -D variants into separate shared objects (here, cross-compiled) and time
mx_mrstft_loss (value + gradient) at BASELINE config 5's size on the GPU box.

    python tools/exp_mrstft.py build  name1:-DFOO name2:-DBAR=1,-DBAZ ...
    python tools/exp_mrstft.py run [B] [T]
    (per-kernel split:  rocprofv3 --kernel-trace --stats -- python3 tools/exp_mrstft.py run)
"""
import ctypes
import glob
import math
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "mod_extraction_amd", "_lib")
SRC = os.path.join(ROOT, "mod_extraction_amd", "csrc")


def build(specs):
    for old in glob.glob(os.path.join(LIB, "expmr_*.so")):
        os.remove(old)
    for spec in specs:
        name, _, flags = spec.partition(":")
        out = os.path.join(LIB, f"expmr_{name}.so")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared",
               "-I", os.path.join(ROOT, "include"), os.path.join(SRC, "mrstft.hip"), "-o", out] + [f for f in flags.split(",") if f]
        print(" ".join(cmd[-3:]), flush=True)
        subprocess.check_call(cmd)


def run(B=256, T=176400):
    import torch
    sys.path.insert(0, ROOT)
    from mod_extraction_amd import mrstft
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    y = (torch.rand(B, T, device=dev) * 2 - 1) * 0.5
    x = (0.8 * y + 0.05 * torch.randn(B, T, device=dev)).clamp(-1, 1)
    mod = mrstft.MultiResolutionSTFTLoss()
    win, tw = mod.buffers_on(dev)
    n_res = 3
    frames = [1 + T // h for h in mod.hop_sizes]
    part = torch.empty(3 * B * max(-(-f // 8) for f in frames), device=dev, dtype=torch.float64)
    coef = torch.empty(n_res, device=dev)
    terms = torch.empty(2 * n_res + 1, device=dev)
    # (generous: variants built with a smaller MR_RUN_MIN keep more run tails than the shipped formula)
    scratch = torch.empty(3 * sum(mrstft.scratch_floats(B, T, n, h) for n, h in zip(mod.fft_sizes, mod.hop_sizes)), device=dev)
    dx = torch.empty((B, T), device=dev)
    ffts = (ctypes.c_int32 * n_res)(*mod.fft_sizes)
    hops = (ctypes.c_int32 * n_res)(*mod.hop_sizes)
    vp = ctypes.c_void_p
    st = vp(torch.cuda.current_stream().cuda_stream)
    for so in sorted(glob.glob(os.path.join(LIB, "expmr_*.so"))) + [os.path.join(LIB, "libmodex_hip.so")]:
        lib = ctypes.CDLL(so)

        def call():
            return lib.mx_mrstft_loss(vp(x.data_ptr()), ctypes.c_int64(T), vp(y.data_ptr()), ctypes.c_int64(T), ctypes.c_int64(B),
                                      ctypes.c_int64(T), ctypes.c_int32(n_res), ffts, hops, vp(win.data_ptr()), vp(tw.data_ptr()),
                                      ctypes.c_float(1.0), ctypes.c_float(1.0), ctypes.c_float(1e-8), vp(part.data_ptr()),
                                      vp(coef.data_ptr()), vp(scratch.data_ptr()), vp(terms.data_ptr()), vp(dx.data_ptr()),
                                      ctypes.c_int64(T), st)
        rc = call()
        assert rc == 0, (so, rc)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            call()
        e1.record()
        torch.cuda.synchronize()
        print(f"{os.path.basename(so):32s} {e0.elapsed_time(e1) / 5:8.3f} ms  loss {float(terms[-1]):.6f}  |dx| {float(dx.abs().sum()):.6e}",
              flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        run(*(int(v) for v in sys.argv[2:4]))
