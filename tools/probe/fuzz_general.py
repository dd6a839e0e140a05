"""GPU box: randomised shapes through the general model paths against the CPU oracle (catches indexing mistakes the fixed test
cases do not reach: kernels taller than the image, dilations beyond it, one-frame clips, strides that do not divide, ...).
    python tools/probe/fuzz_general.py [n_cases] [seed]
"""
import os, sys, random
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from oracle import models as om, tcn_general as ot
from mod_extraction_amd import models as am, tcn as at

dev = torch.device("cuda:0")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-20))


def loss(t):
    return (t * torch.linspace(0.5, 1.5, t.numel(), device=t.device).view_as(t)).sum() / t.numel()


worst = {"cnn_out": 0.0, "cnn_grad": 0.0, "tcn_out": 0.0, "tcn_grad": 0.0, "lstm_out": 0.0, "lstm_grad": 0.0}
for case in range(n_cases):
    torch.manual_seed(1000 + case)
    # ---- Spectral2DCNN
    nb = rng.randint(1, 3)
    pool = rng.choice([1, 2, 3])
    n_mels = rng.choice([8, 12, 20, 27, 33]) * (pool ** nb) // rng.choice([1, 1, 2]) or pool ** nb
    n_mels = max(n_mels, pool ** nb)
    cfg = dict(in_ch=rng.randint(1, 3), n_samples=rng.choice([1500, 4000, 9000]), n_fft=rng.choice([512, 1024]), hop_len=rng.choice([128, 256, 300]),
               n_mels=min(n_mels, 120), kernel_size=(rng.randint(1, 6), rng.randint(1, 9)), out_channels=[rng.randint(1, 9) for _ in range(nb)],
               bin_dilations=[rng.randint(1, 4) for _ in range(nb)], temp_dilations=[rng.randint(1, 20) for _ in range(nb)],
               pool_size=(pool, 1), latent_dim=rng.randint(1, 3), use_ln=rng.random() < 0.6)
    if cfg["n_mels"] < pool ** nb:
        cfg["n_mels"] = pool ** nb
    ref = om.Spectral2DCNN(**cfg)
    mine = am.Spectral2DCNN(**cfg)
    mine.load_state_dict(ref.state_dict()); mine = mine.to(dev); ref.eval(); mine.eval()
    if mine.generic:
        x = torch.rand(2, cfg["in_ch"], cfg["n_samples"]) * 2 - 1
        o_m, l_m = mine(x.to(dev)); (loss(o_m) + 0.1 * loss(l_m)).backward()
        o_r, l_r = ref(x); (loss(o_r) + 0.1 * loss(l_r)).backward()
        worst["cnn_out"] = max(worst["cnn_out"], rel(o_m.detach().cpu(), o_r.detach()), rel(l_m.detach().cpu(), l_r.detach()))
        gmax = max(float(p.grad.abs().max()) for p in ref.parameters())
        for (n, p), q in zip(mine.named_parameters(), ref.parameters()):
            if float(q.grad.abs().max()) > 1e-4 * gmax:
                e = rel(p.grad.cpu(), q.grad)
                if e > 3e-5:                                  # arbitrate in fp64: which side is off?
                    import copy
                    r64 = copy.deepcopy(ref).double(); r64.zero_grad()
                    o64, l64 = r64(x.double()); (loss(o64) + 0.1 * loss(l64)).backward()
                    g64 = dict(r64.named_parameters())[n].grad
                    e_m, e_r = rel(p.grad.cpu().double(), g64), rel(q.grad.double(), g64)
                    print(f"CNN arbitration {n}: device-vs-oracle {e:.2e}; vs fp64: device {e_m:.2e}, fp32 oracle {e_r:.2e}")
                    if e_m > max(2e-5, 3 * e_r):
                        print("CNN MISMATCH", cfg, n, e_m, e_r)
                    e = min(e, e_m)
                worst["cnn_grad"] = max(worst["cnn_grad"], e)
    # ---- TCN
    nbt = rng.randint(1, 3)
    causal = rng.random() < 0.5
    k = rng.randint(1, 7)
    dil = [rng.randint(1, 5) for _ in range(nbt)]
    strides = [rng.randint(1, 3) for _ in range(nbt)]
    use_res = rng.random() < 0.7 and (causal or k % 2 == 1)
    pad = 0 if causal else (None if (k % 2 == 1 and rng.random() < 0.5) else rng.randint(0, (min(dil) * (k - 1)) // 2))
    cond_dim = rng.choice([0, 0, 3])
    kw = dict(out_channels=[rng.randint(1, 10) for _ in range(nbt)], dilations=dil, in_ch=rng.randint(1, 6), kernel_size=k, strides=strides,
              padding=pad, use_ln=False, use_act=rng.random() < 0.8, use_res=use_res, cond_dim=cond_dim, use_film_bn=rng.random() < 0.5,
              is_causal=causal, is_cached=False)
    T = rng.randint(40, 500)
    try:
        tref = ot.TCN(**kw)
        xr = torch.randn(2, kw["in_ch"], T, requires_grad=True)
        cr = torch.randn(2, cond_dim, requires_grad=True) if cond_dim else None
        yr = tref(xr, cr)
    except (AssertionError, RuntimeError) as ex:        # a residual that cannot be cropped etc.: the reference rejects it too
        yr = None
    if yr is not None and yr.size(-1) >= 1:
        tmine = at.TCN(**kw); tmine.load_state_dict(tref.state_dict()); tmine = tmine.to(dev)
        xm = xr.detach().to(dev).requires_grad_(True)
        cm = cr.detach().to(dev).requires_grad_(True) if cond_dim else None
        ym = tmine(xm, cm)
        assert ym.shape == yr.shape, (kw, T, ym.shape, yr.shape)
        loss(yr).backward(); loss(ym).backward()
        worst["tcn_out"] = max(worst["tcn_out"], rel(ym.detach().cpu(), yr.detach()))
        e = rel(xm.grad.cpu(), xr.grad)
        gmax = max(float(p.grad.abs().max()) for p in tref.parameters())
        for (n, p), q in zip(tmine.named_parameters(), tref.parameters()):
            if float(q.grad.abs().max()) > 1e-4 * gmax:
                e = max(e, rel(p.grad.cpu(), q.grad))
        if e > 1e-3:
            print("TCN MISMATCH", kw, T, e)
        worst["tcn_grad"] = max(worst["tcn_grad"], e)
    # ---- LSTM
    ch = rng.choice([(1, 1), (2, 2), (1, 3), (3, 1)])
    args = (ch[0], ch[1], rng.randint(1, 140), rng.randint(1, 4))
    lref, lmine = om.LSTMEffectModel(*args), am.LSTMEffectModel(*args)
    lmine.load_state_dict(lref.state_dict()); lmine = lmine.to(dev)
    if lmine.generic:
        Tn = rng.randint(1, 200)
        x = torch.rand(2, args[0], Tn) - 0.5
        lat_r = torch.rand(2, args[3], Tn, requires_grad=True)
        lat_m = lat_r.detach().to(dev).requires_grad_(True)
        yr = lref(x, lat_r); ym = lmine(x.to(dev), lat_m)
        loss(yr).backward(); loss(ym).backward()
        worst["lstm_out"] = max(worst["lstm_out"], float((ym.detach().cpu() - yr.detach()).abs().max()))
        e = rel(lat_m.grad.cpu(), lat_r.grad)
        for p, q in zip(lmine.parameters(), lref.parameters()):
            e = max(e, rel(p.grad.cpu(), q.grad))
        if e > 1e-3:
            print("LSTM MISMATCH", args, Tn, e)
        worst["lstm_grad"] = max(worst["lstm_grad"], e)
print("cases", n_cases, {k: f"{v:.2e}" for k, v in worst.items()})
