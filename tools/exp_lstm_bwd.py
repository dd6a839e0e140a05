"""Variant harness for the LSTM backward kernel (csrc/lstm.hip): build -D variants here, time them on the GPU box.

Round 5 record (the variants were a -DLSB_VAR switch in lstm_bwd_kernel, removed again): the bare backward step
(mx_lstm_step_probe(1)) takes 279 ns, the kernel 468 ns per step; where do its two fp32 matrix instructions per step pair
(dW_hh, 77 ns per step in round 4) belong?  One box, 128 clips x 1024 steps, ns per step:
    round-4 placement (both at the top of the pair's step, behind their operand reads)                      468
    one per step: the first behind the packed FMAs (operands read with the gate gradients), the second at
      the top of the next step from registers                                                                593
    both behind the packed FMAs of the pair's step                                                           546
    one per step, both behind the packed FMAs                                                                610
    none (restructured loop, wrong results)                                                                  451   (round-4 loop without them: 402)
i.e. v_mfma_f32_32x32x2_f32 does not run beside the wave's (or the SIMD's other wave's) vector stream: every placement inside
the dependent chain costs MORE than the 64-cycle pipe time of the instruction, and the round-4 placement -- at the step top,
where the waves re-converge from the barrier anyway -- is the cheapest found.  The weight-gradient product is 16 384 MAC per
step and clip = 128 cycles of fp32 matrix pipe per SIMD and step (61 ns): that part of the gap to the bare floor is not
schedule, it is the price of exact-fp32 weight gradients inside the recurrence (a separate product kernel: 86 us per chunk,
DESIGN section 8.1).
    python tools/exp_lstm_bwd.py build v1:-DLSB_VAR=1 v2:-DLSB_VAR=2 ...     (CPU container)
    python tools/exp_lstm_bwd.py run                                          (GPU box; every _lib/explstm_*.so + the product build)
"""
import glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "mod_extraction_amd", "_lib")
SRC = os.path.join(ROOT, "mod_extraction_amd", "csrc")


def build(specs):
    for old in glob.glob(os.path.join(LIB, "explstm_*.so")):
        os.remove(old)
    objs = [o for o in glob.glob(os.path.join(LIB, "obj", "*.o")) if not o.endswith("/lstm.o")]
    for spec in specs:
        name, _, flags = spec.partition(":")
        obj = f"/tmp/explstm_{name}.o"
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                               "-fvisibility=hidden", "-I", os.path.join(ROOT, "include"), "-c", os.path.join(SRC, "lstm.hip"), "-o", obj]
                              + [f for f in flags.split(",") if f])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                               os.path.join(LIB, f"explstm_{name}.so")] + objs + [obj])
        print("built", name, flags, flush=True)


def run():
    for so in [None] + sorted(glob.glob(os.path.join(LIB, "explstm_*.so"))):
        env = dict(os.environ)
        if so:
            env["MODEX_HIP_LIB"] = so
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_lstm.py")], env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        d = json.loads(line[-1]) if line else {"error": out.stderr[-300:]}
        print(f"{os.path.basename(so)[8:-3] if so else 'product':12s}", {k: d.get(k) for k in ("fwd_stash_ms", "bwd_total_ms", "bwd_ns_per_step", "error") if k in d}, flush=True)


if __name__ == "__main__":
    build(sys.argv[2:]) if sys.argv[1] == "build" else run()
