"""Which XCDs / CUs does a CU-masked stream use?  (KFD may spread mask bits round-robin over the XCCs.)
    hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/probe/xcc_probe.so tools/probe/xcc_probe.hip
    gpurun -- 'python tools/probe/probe_xcc.py'
Prints, per mask, the histogram of XCC_ID over 2048 resident workgroups and the number of distinct (xcc, se, cu)."""
import collections, ctypes, os, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(here, "..", ".."))
from mod_extraction_amd import streams
lib = ctypes.CDLL(os.path.join(here, "xcc_probe.so"))
dev = torch.device("cuda:0")
hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
n_cu = torch.cuda.get_device_properties(0).multi_processor_count
words = n_cu // 32
print("CUs", n_cu, "mask words", words)
F = 0xFFFFFFFF


def run(name, mask):
    st = streams._masked_stream(hip, mask, dev) if mask is not None else torch.cuda.current_stream()
    out = torch.zeros(2048 * 2, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    rc = lib.where(ctypes.c_void_p(out.data_ptr()), 2048, 200000, ctypes.c_void_p(st.cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    o = out.cpu().view(-1, 2).tolist()
    xcc = collections.Counter(x & 0xF for x, _ in o)
    # HW_ID (gfx9): wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13(+), ...
    cus = {(x & 0xF, (h >> 13) & 0x7, (h >> 12) & 1, (h >> 8) & 0xF) for x, h in o}
    print(f"{name:34s} xcc histogram {dict(sorted(xcc.items()))}  distinct (xcc,se,sh,cu): {len(cus)}")


run("no mask", None)
run("word 0 only", [F] + [0] * (words - 1))
run("words 0-4 (streams.py main)", [F] * 5 + [0] * 3)
run("words 5-7 (streams.py side)", [0] * 5 + [F] * 3)
run("bits = 0 mod 8 (every 8th bit)", [0x01010101] * words)
run("bits 0-4 mod 8", [0x1F1F1F1F] * words)
run("bits 5-7 mod 8", [0xE0E0E0E0] * words)
