"""Kernel experiment harness for the sparse gradient kernels: build -D variants of wgrad_sp_f16.hip / dgrad_sp_f16.hip into
separate shared objects (CPU container) and time the entry points on the GPU box with realistic operands.

    python tools/exp_sparse.py build [--clean] name1:-DFOO name2:-DBAR=1,-DBAZ ...     (--clean removes earlier variants)
    python tools/exp_sparse.py run [B]
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
    if specs and specs[0] == "--clean":
        for old in glob.glob(os.path.join(LIB, "exps_*.so")):
            os.remove(old)
        specs = specs[1:]
    for spec in specs:
        name, _, flags = spec.partition(":")
        out = os.path.join(LIB, f"exps_{name}.so")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared",
               "-I", os.path.join(ROOT, "include"), os.path.join(SRC, "wgrad_sp_f16.hip"), os.path.join(SRC, "dgrad_sp_f16.hip"),
               "-o", out] + [f for f in flags.split(",") if f]
        print(" ".join(cmd[-3:]), flush=True)
        subprocess.check_call(cmd)


def run(B=64):
    import torch
    dev = torch.device("cuda:0")
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
    only = os.environ.get("EXP_ONLY", "wgrad,dgrad").split(",")
    shapes = [tuple(int(v) for v in s.split("x")) for s in os.environ.get("EXP_SHAPES", "64x1,32x2,16x4").split(",")]
    for so in sorted(glob.glob(os.path.join(LIB, "exps_*.so"))):
        lib = ctypes.CDLL(so)
        name = os.path.basename(so)[5:-3]
        for (H, T) in shapes:
            Hp = H // 2
            g = torch.Generator(device="cpu").manual_seed(0)
            x = torch.randn((B, H, 4, 352, 16), generator=g).to(dev)
            x[:, :, :, 345:, :] = 0
            x_hi = x.half()
            x_lo = (x - x_hi.float()).half()
            G = torch.randn((B, 64, Hp, 352), generator=g).to(dev)
            G[..., 345:] = 0
            amax = torch.randint(0, 2, (B, 64, Hp, 352), generator=g, dtype=torch.uint8).to(dev)
            scale = torch.tensor([256.0, 1 / 256.0], device=dev)
            W = torch.randn((64, 64, 5, 13), generator=g).to(dev) * 0.05
            st = vp(torch.cuda.current_stream().cuda_stream)
            gidx = torch.empty((B, 64, Hp, 22, 2), device=dev, dtype=torch.int16)
            gc_hi = torch.empty((B, Hp, 4, 352, 16), device=dev, dtype=torch.half)
            gc_lo = torch.empty_like(gc_hi)
            gc_idx = torch.empty((B, Hp, 4, 352), device=dev, dtype=torch.int32)
            rc = lib.mx_conv_prep_gpool_cl_f16(vp(G.data_ptr()), vp(amax.data_ptr()), vp(scale.data_ptr()), i64(B), i64(H), i64(345),
                                               vp(gc_hi.data_ptr()), vp(gc_lo.data_ptr()), vp(gc_idx.data_ptr()),
                                               vp(gidx.data_ptr()), st)
            assert rc == 0, rc
            w_hi = torch.empty(4 * 3 * 2 * 13 * 2 * 64 * 16, device=dev, dtype=torch.half)
            w_lo = torch.empty_like(w_hi)
            assert lib.mx_conv_pack_weights_sp_f16(vp(W.data_ptr()), vp(w_hi.data_ptr()), vp(w_lo.data_ptr()), st) == 0
            rps = max(1, (B * Hp) // 256)
            n_slabs = -(-(B * Hp) // rps)
            part = torch.empty(n_slabs * 65 * 64 * 64, device=dev)
            dW = torch.empty(64 * 64 * 65, device=dev)
            dx = torch.empty((B, 64, H, 352), device=dev)

            def wgr():
                return lib.mx_conv_block_wgrad_sp_f16(vp(gc_hi.data_ptr()), vp(gc_lo.data_ptr()), vp(gidx.data_ptr()),
                                                      vp(x_hi.data_ptr()), vp(x_lo.data_ptr()), vp(scale.data_ptr()), i64(B), i64(H),
                                                      i64(345), i32(T), i64(rps), vp(part.data_ptr()), vp(dW.data_ptr()), st)

            ln_part = torch.empty((B, 64, H, 2, 2), device=dev)
            ln_args = (vp(x_hi.data_ptr()), vp(x_lo.data_ptr()), vp(ln_part.data_ptr())) if os.environ.get('EXP_LN') else (vp(0), vp(0), vp(0))

            def dgr():
                return lib.mx_conv_block_dgrad_sp_f16(vp(gc_hi.data_ptr()), vp(gc_lo.data_ptr()), vp(gc_idx.data_ptr()),
                                                      vp(w_hi.data_ptr()), vp(w_lo.data_ptr()), vp(scale.data_ptr()), i64(B), i64(H),
                                                      i64(345), i32(T), vp(dx.data_ptr()), *ln_args, vp(0), st)
            for fn, tag in ((wgr, "wgrad"), (dgr, "dgrad")):
                if tag not in only:
                    continue
                rc = fn()
                assert rc == 0, (name, tag, rc)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for _ in range(3):
                    fn()
                e0.record()
                for _ in range(10):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 10
                chk = float(dW[1234]) if tag == "wgrad" else float(dx.flatten()[123456])
                print(f"{name:20s} H={H:4d} T={T} {tag:6s} {ms:8.3f} ms  chk={chk:.6g}", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 64)
