"""Derives the operand layouts of v_smfmac_f32_16x16x64_f16 empirically (one-hot compressed A elements x coded B), as
probe_smfmac.py does for the 32x32x32 shape.  Prints, per (lane, element j, selector f): the D row that lit up and which B
(lane, element) fed each of the 16 columns -- i.e. the logical k the compressed element stands for."""
import ctypes, os
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "smfmac_probe.so"))
dev = torch.device("cuda:0")
probes = [(la, j, f) for la in range(64) for j in range(8) for f in range(4)]
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
d = torch.zeros((P, 64, 4), dtype=torch.float32, device=dev)
vp = ctypes.c_void_p
rc = lib.probe16(vp(a.data_ptr()), vp(b.data_ptr()), vp(idx.data_ptr()), vp(d.data_ptr()), P, 0)
assert rc == 0
d = d.cpu().numpy()
# D layout assumption (16x16 MFMA): lane l, reg r -> row 4 (l >> 4) + r, col l & 15
D = np.zeros((P, 16, 16), dtype=np.float32)
for l in range(64):
    for r in range(4):
        D[:, 4 * (l >> 4) + r, l & 15] = d[:, l, r]
res = {}
for p, (la, j, f) in enumerate(probes):
    rows = np.nonzero(np.abs(D[p]).sum(axis=1))[0]
    if len(rows) != 1:
        res[(la, j, f)] = ("rows", rows.tolist()); continue
    m = int(rows[0])
    codes = D[p, m].astype(np.int64) - 1
    lb, jb = codes // 16, codes % 16               # B lane / element that fed column n
    res[(la, j, f)] = (m, lb.tolist(), jb.tolist())
for la in (0, 1, 15, 16, 17, 31, 32, 47, 48, 63):
    for j in range(8):
        print(la, j, [(res[(la, j, f)][0], res[(la, j, f)][1][:3], sorted(set(res[(la, j, f)][2]))) for f in range(4)])
# hypothesis check: row m = la & 15; B lane for column n = n + 16 * s
ok_row = all(isinstance(v[0], int) and v[0] == (k[0] & 15) for k, v in res.items())
print("row == lane & 15 for all:", ok_row)
# derive logical k assuming B: lane l -> column l & 15, k = 16 (l >> 4) + jb   (to be confirmed by the pattern above)
tab = {}
for (la, j, f), v in res.items():
    if not isinstance(v[0], int):
        continue
    s = sorted(set((np.array(v[1]) - np.arange(16)) // 16))
    jb = sorted(set(v[2]))
    tab[(la >> 4, j, f)] = tab.get((la >> 4, j, f), set()) | {(tuple(s), tuple(jb))}
for key in sorted(tab):
    print("kgroup", key[0], "elem", key[1], "sel", key[2], "->", sorted(tab[key]))
