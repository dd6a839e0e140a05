"""Numeric check of the derived v_smfmac_f32_32x32x32_f16 semantics:
   A: lane la -> row m = la & 31, half hh = la >> 5; compressed element j (0..7) sits in logical group
      kbase = 16 * (j >> 2) + 8 * hh + 4 * ((j >> 1) & 1) at position kbase + f_j, f_j = (idx >> 2j) & 3
   B: lane lb -> col n = lb & 31, logical k = 16 * (lb >> 5) + jb
   D: lane l, reg r -> row (r&3) + 8(r>>2) + 4(l>>5), col l & 31"""
import ctypes, os
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "smfmac_probe.so"))
dev = torch.device("cuda:0")
rng = np.random.RandomState(0)
P = 8
A_dense = np.zeros((P, 32, 32), dtype=np.float32)
a = np.zeros((P, 64, 8), dtype=np.float16)
idx = np.zeros((P, 64), dtype=np.int32)
for p in range(P):
    for la in range(64):
        m, hh = la & 31, la >> 5
        word = 0
        for gq in range(4):                      # 4 groups of 4 logical k per lane
            kbase = 16 * (gq >> 1) + 8 * hh + 4 * (gq & 1)
            if p % 2 == 0:                       # our pattern: one nonzero in {0,1}, one in {2,3}
                f0, f1 = rng.randint(0, 2), 2 + rng.randint(0, 2)
            else:                                # arbitrary increasing pair
                f0, f1 = sorted(rng.choice(4, 2, replace=False))
            v0, v1 = rng.uniform(-1, 1), rng.uniform(-1, 1)
            a[p, la, 2 * gq], a[p, la, 2 * gq + 1] = v0, v1
            A_dense[p, m, kbase + f0] = np.float16(v0)
            A_dense[p, m, kbase + f1] = np.float16(v1)
            word |= (f0 << (4 * gq)) | (f1 << (4 * gq + 2))
        idx[p, la] = word
Bm = rng.uniform(-1, 1, size=(P, 32, 32)).astype(np.float16)        # [k][n]
b = np.zeros((P, 64, 16), dtype=np.float16)
for lb in range(64):
    for jb in range(16):
        b[:, lb, jb] = Bm[:, 16 * (lb >> 5) + jb, lb & 31]
ta, tb, ti = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev), torch.from_numpy(idx).to(dev)
d = torch.zeros((P, 64, 16), dtype=torch.float32, device=dev)
vp = ctypes.c_void_p
assert lib.probe(vp(ta.data_ptr()), vp(tb.data_ptr()), vp(ti.data_ptr()), vp(d.data_ptr()), P, 1) == 0
d = d.cpu().numpy()
D = np.zeros((P, 32, 32), dtype=np.float32)
for l in range(64):
    for r in range(16):
        D[:, (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31] = d[:, l, r]
ref = np.einsum("pmk,pkn->pmn", A_dense.astype(np.float64), Bm.astype(np.float64))
print("max abs err", np.abs(D - ref).max(), "max ref", np.abs(ref).max())
