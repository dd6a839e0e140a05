"""Derives the operand layouts of v_smfmac_f32_32x32x32_f16 empirically (one-hot A elements x coded B)."""
import ctypes, os, sys
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "smfmac_probe.so"))
dev = torch.device("cuda:0")
lanes = list(range(64))
probes = [(la, j, f) for la in lanes for j in range(8) for f in range(4)]
P = len(probes)
a = torch.zeros((P, 64, 8), dtype=torch.float16)
idx = torch.zeros((P, 64), dtype=torch.int32)
for p, (la, j, f) in enumerate(probes):
    a[p, la, j] = 1.0
    idx[p, la] = f << (2 * j)
b = torch.zeros((64, 16), dtype=torch.float16)
for l in range(64):
    for j in range(16):
        b[l, j] = float(l * 16 + j + 1)          # 1..1024, exact in fp16
a, b, idx = a.to(dev), b.to(dev), idx.to(dev)
d = torch.zeros((P, 64, 16), dtype=torch.float32, device=dev)
vp = ctypes.c_void_p
rc = lib.probe(vp(a.data_ptr()), vp(b.data_ptr()), vp(idx.data_ptr()), vp(d.data_ptr()), P, 0)
assert rc == 0
d = d.cpu().numpy()
# D layout assumption (32x32 MFMA): lane l, reg r -> row (r&3)+8(r>>2)+4(l>>5), col l&31
D = np.zeros((P, 32, 32), dtype=np.float32)
for l in range(64):
    for r in range(16):
        D[:, (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31] = d[:, l, r]
out = []
for p, (la, j, f) in enumerate(probes):
    rows = np.nonzero(np.abs(D[p]).sum(axis=1))[0]
    if len(rows) == 0:
        out.append((la, j, f, None, None)); continue
    m = int(rows[0])
    codes = D[p, m].astype(np.int64) - 1
    lb, jb = codes // 16, codes % 16
    # expect lb = n + 32 * s for some s, jb constant
    s_set = sorted(set(((lb - np.arange(32)) // 32).tolist())); j_set = sorted(set(jb.tolist()))
    out.append((la, j, f, (rows.tolist() if len(rows) > 1 else m), (s_set, j_set)))
for la in (0, 1, 31, 32, 33, 63):
    for j in range(8):
        print(la, j, [(o[3], o[4]) for o in out if o[0] == la and o[1] == j])
# consistency summary: m == la & 31 ?
bad = [o for o in out if o[3] is not None and o[3] != (o[0] & 31)]
print("rows != lane&31:", len(bad), bad[:5])
