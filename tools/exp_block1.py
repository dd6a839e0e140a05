"""Ablation harness for the first block's forward kernel (conv1_f16x3_persist_kernel in conv_f16.hip): build -D variants
here, time mx_conv_block1_fwd_f16 on the GPU box with random operands at the headline size.

    python tools/exp_block1.py build name1:-DC1_ABL=1 name2:...      (CPU container)
    python tools/exp_block1.py run [B]                               (GPU box; every _lib/expb1_*.so)
C1_ABL bits (wrong results): 1 = no epilogue, 2 = no matrix instructions, 4 = epilogue without its global stores,
8 = no LDS-DMA inside the loop (the prologue's patch is reused).
"""
import ctypes
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "mod_extraction_amd", "_lib")
SRC = os.path.join(ROOT, "mod_extraction_amd", "csrc")


def build(specs):
    for old in glob.glob(os.path.join(LIB, "expb1_*.so")):
        os.remove(old)
    for spec in specs:
        name, _, flags = spec.partition(":")
        out = os.path.join(LIB, f"expb1_{name}.so")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared",
               "-I", os.path.join(ROOT, "include"), os.path.join(SRC, "conv_f16.hip"), "-o", out] + [f for f in flags.split(",") if f]
        print(" ".join(cmd[-3:]), flush=True)
        subprocess.check_call(cmd)


def run(B=256, H=256):
    import torch
    dev = torch.device("cuda:0")
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn((B, H, 352, 16), generator=g).to(dev)
    x[:, :, 345:, :] = 0
    x[..., 10:] = 0
    x_hi = x.half()
    x_lo = (x - x_hi.float()).half()
    w = (torch.randn((2, 13 * 1024), generator=g) * 8).to(dev)
    w_hi = w.half()
    w_lo = (w - w_hi.float()).half()
    bias = torch.randn(64, generator=g).to(dev)
    slope = torch.full((64,), 0.25, device=dev)
    out = torch.empty((B, 64, H // 2, 352), device=dev)
    amax = torch.empty((B, 64, H // 2, 352), device=dev, dtype=torch.uint8)
    part = torch.empty((B, H // 2, 64, 2), device=dev)
    st = vp(torch.cuda.current_stream().cuda_stream)
    for so in sorted(glob.glob(os.path.join(LIB, "expb1_*.so"))):
        lib = ctypes.CDLL(so)
        name = os.path.basename(so)[6:-3]
        # the library reads the variable once (static): a variant whose name ends in "p1" runs the row-exchanging layout
        os.environ["MODEX_BLOCK1_PERSIST"] = "1" if name.endswith("p1") else "3" if name.endswith("p3") else "2"

        def call():
            return lib.mx_conv_block1_fwd_f16(vp(x_hi.data_ptr()), vp(x_lo.data_ptr()), vp(w_hi[0].data_ptr()),
                                              vp(w_lo[0].data_ptr()), vp(bias.data_ptr()), i64(B), i64(H), i64(345),
                                              vp(out.data_ptr()), vp(amax.data_ptr()), vp(slope.data_ptr()),
                                              vp(part.data_ptr()), st)
        rc = call()
        assert rc == 0, rc
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        times = []
        for _ in range(3):
            e0.record()
            for _ in range(10):
                call()
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) / 10)
        if hasattr(lib, "mx_diag_c1"):
            buf = (ctypes.c_uint64 * 8)()
            lib.mx_diag_c1(buf, 1)
            call()
            torch.cuda.synchronize()
            lib.mx_diag_c1(buf, 0)
            n = B * (H // 2)
            print("   cycles per row pair (wave 0): head %.0f  taps %.0f  dma wait %.0f  epilogue (rest) %.0f | head barrier %.0f  five tiles %.0f  "
                  "mid barrier %.0f  mid tile %.0f" % tuple(buf[i] / n for i in range(8)))
        print(f"{name:24s} persist={os.environ['MODEX_BLOCK1_PERSIST']} {min(times):8.3f} ms  chk={float(out.float().abs().mean()):.6g}", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 256)
