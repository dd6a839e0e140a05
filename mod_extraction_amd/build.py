"""Build libmodex_hip.so (all HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python -m mod_extraction_amd.build [--force]

hipcc cross-compiles without a GPU.  Each .hip file is compiled to an object (cached by mtime),
then linked into mod_extraction_amd/_lib/libmodex_hip.so, which travels to the GPU box with the
repo snapshot.  -ffp-contract=off: the bit-exact kernels (LFO phase, flanger indices, corner
bookkeeping) must round where the reference's separate torch ops round; FILE_FLAGS turns contraction back on for the
tolerance-tested FFT kernels.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "_lib")
OBJ_DIR = os.path.join(OUT_DIR, "obj")
SO = os.path.join(OUT_DIR, "libmodex_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fvisibility=hidden", "-Wall", "-Wno-unused-function", "-I", os.path.join(HERE, "..", "include")]


# Per-file overrides.  The FFT kernels (MR-STFT loss, log-mel front end) are VALU-issue bound and only tolerance-tested
# (1e-5 against torch.stft-based oracles): with contraction a twiddle multiply is 2 mul + 2 fma instead of 4 mul + 2 add.
# Everything with a bit-exact contract (lfo, flanger, corners, phaser's JUCE-order kernel, ...) keeps -ffp-contract=off.
# (mrstft.hip stays off and writes its twiddle FMAs by hand: see cmulf there -- blanket contraction breaks the exact
# x == y symmetry of the packed two-signal transform.)
FILE_FLAGS = {"melspec.hip": ["-ffp-contract=fast"]}


def _newer(a: str, b: str) -> bool:
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def sources():
    return sorted(f for f in os.listdir(SRC) if f.endswith(".hip"))


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ_DIR, exist_ok=True)
    headers = [os.path.join(SRC, f) for f in os.listdir(SRC) if f.endswith(".h")]
    hdr_time = max([os.path.getmtime(h) for h in headers] + [0.0])
    jobs = []
    for f in sources():
        src, obj = os.path.join(SRC, f), os.path.join(OBJ_DIR, f[:-4] + ".o")
        stale = force or _newer(src, obj) or (os.path.exists(obj) and hdr_time > os.path.getmtime(obj))
        if stale:
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [HIPCC] + FLAGS + FILE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        if verbose:
            print("[build]", os.path.basename(src), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(OBJ_DIR, f[:-4] + ".o") for f in sources()]
    if force or jobs or not os.path.exists(SO):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs
        subprocess.check_call(cmd)
        if verbose:
            print("[build] linked", SO, flush=True)
    return SO


if __name__ == "__main__":
    build(force="--force" in sys.argv)
