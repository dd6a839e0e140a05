"""Kernel experiment harness for one .hip file: build variants with -D flags into separate shared objects (here, by
cross-compilation) and time the f16x3 conv entry points on the GPU box.

    python tools/exp_conv_f16.py build  name1:-DFOO name2:-DBAR=1 ...     (CPU container; add the #ifdef you want to test)
    python tools/exp_conv_f16.py run [B]                                   (GPU box; every _lib/exp_*.so)
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
    for old in glob.glob(os.path.join(LIB, "exp_*.so")):
        os.remove(old)
    for spec in specs:
        name, _, flags = spec.partition(":")
        out = os.path.join(LIB, f"exp_{name}.so")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared",
               "-I", os.path.join(ROOT, "include"), os.path.join(SRC, "conv_f16.hip"), os.path.join(SRC, "wgrad_f16.hip"),
               "-o", out] + [f for f in flags.split(",") if f]
        print(" ".join(cmd[-4:]), flush=True)
        subprocess.check_call(cmd)


def run(B=64):
    import torch
    dev = torch.device("cuda:0")
    vp = ctypes.c_void_p
    res = {}
    for so in sorted(glob.glob(os.path.join(LIB, "exp_*.so"))):
        lib = ctypes.CDLL(so)
        name = os.path.basename(so)[4:-3]
        for (H, T) in ((128, 1), (64, 1), (32, 2)):
            g = torch.Generator(device="cpu").manual_seed(0)
            x = torch.randn((B, H, 4, 352, 16), generator=g).to(dev)
            x_hi = x.half()
            x_lo = (x - x_hi.float()).half()
            w = (torch.randn((4 * 5 * 13 * 64 * 16,), generator=g) * 8).to(dev)
            w_hi = w.half()
            w_lo = (w - w_hi.float()).half()
            bias = torch.zeros(64, device=dev)
            scale = torch.ones(2, device=dev)
            out_p = torch.empty((B, 64, H // 2, 352), device=dev)
            am = torch.empty((B, 64, H // 2, 352), device=dev, dtype=torch.uint8)
            out_d = torch.empty((B, 64, H, 352), device=dev)
            st = vp(torch.cuda.current_stream().cuda_stream)
            i64, i32 = ctypes.c_int64, ctypes.c_int32

            def fwd():
                return lib.mx_conv_block_fwd_f16(vp(x_hi.data_ptr()), vp(x_lo.data_ptr()), vp(w_hi.data_ptr()),
                                                 vp(w_lo.data_ptr()), vp(bias.data_ptr()), i64(B), i64(H), i64(345), i32(T),
                                                 vp(out_p.data_ptr()), vp(am.data_ptr()), None, None, st)

            def dgr():
                return lib.mx_conv_block_dgrad_f16(vp(x_hi.data_ptr()), vp(x_lo.data_ptr()), vp(w_hi.data_ptr()),
                                                   vp(w_lo.data_ptr()), vp(scale.data_ptr()), i64(B), i64(H), i64(345), i32(T),
                                                   vp(out_d.data_ptr()), st)
            rps = max(1, -(-(B * H) // 408))
            n_slabs = -(-(B * H) // rps)
            part = torch.empty(n_slabs * 65 * 64 * 64, device=dev)
            dW = torch.empty(64 * 64 * 65, device=dev)

            def wgr():
                return lib.mx_conv_block_wgrad_f16(vp(x_hi.data_ptr()), vp(x_lo.data_ptr()), vp(x_hi.data_ptr()),
                                                   vp(x_lo.data_ptr()), vp(scale.data_ptr()), i64(B), i64(H), i32(T), i64(rps),
                                                   vp(part.data_ptr()), vp(dW.data_ptr()), st)
            only = os.environ.get("EXP_ONLY", "fwd,dgrad,wgrad").split(",")
            for fn, tag in ((fwd, "fwd"), (dgr, "dgrad"), (wgr, "wgrad")):
                if tag not in only:
                    continue
                rc = fn()
                assert rc == 0, (name, tag, rc)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 5
                tf = 2.0 * 64 * 64 * 65 * B * H * 345 / ms / 1e9
                res[(name, H, T, tag)] = ms
                print(f"{name:24s} H={H:4d} T={T} {tag:6s} {ms:8.3f} ms  {tf:7.1f} TF(alg)  chk={float(out_d.flatten()[12345]) if tag == 'dgrad' else (float(dW[1234]) if tag == 'wgrad' else float(out_p.flatten()[12345])):.5g}",
                      flush=True)
    return res


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 64)
