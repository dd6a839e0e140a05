"""Deterministic little audio corpus for the file-backed dataset tests (written into a temp directory by both the
golden generator and the tests; nothing binary is committed).  Covers: mono int16, stereo float32 with a silent
stretch, a file that is too short, a file with the wrong sample rate, a mostly-silent file that forces the
retry / next-file fall-back, nested directories, dot-files; a wet directory with the same names; two pre-rendered
dataset directories in the `<stem>.pt + <stem>_dry.wav + <stem>_wet.wav` format."""
import os

import numpy as np

SR = 44100
N = 4410


def _write(path, data, sr=SR):
    from scipy.io import wavfile
    os.makedirs(os.path.dirname(path), exist_ok=True)
    wavfile.write(path, sr, data)


def make_corpus(root: str) -> dict:
    rng = np.random.RandomState(1234)
    dry, wet = os.path.join(root, "dry"), os.path.join(root, "wet")

    def noise(n, ch=1, amp=0.5):
        x = rng.uniform(-amp, amp, size=(n, ch)).astype(np.float32)
        return x[:, 0] if ch == 1 else x

    files = {}
    a = (noise(SR) * 32767).astype(np.int16)                                  # mono int16, 1 s
    b = noise(2 * SR, ch=2)                                                   # stereo float32, silent middle
    b[30000:52000, :] = 0.0
    c = noise(N // 2)                                                         # too short
    d = noise(SR)                                                             # wrong rate
    e = noise(SR, amp=1e-4)                                                   # silent almost everywhere ...
    e[20000:20000 + N + 600] = noise(N + 600)                                 # ... except one island
    f = noise(3 * SR // 2)
    for name, data, sr in (("a.wav", a, SR), ("sub/b.wav", b, SR), ("c.wav", c, SR), ("d.wav", d, 22050),
                           ("e.wav", e, SR), ("sub/deeper/f.wav", f, SR)):
        _write(os.path.join(dry, name), data, sr)
        w = data.astype(np.float32) / 32768.0 if data.dtype == np.int16 else data
        _write(os.path.join(wet, name), (np.tanh(2.0 * w)).astype(np.float32), sr)
        files[name] = data
    _write(os.path.join(dry, ".hidden.wav"), noise(SR))
    # pre-rendered datasets
    import torch
    for tag, count in (("pre_a", 3), ("pre_b", 3)):
        out = os.path.join(root, tag)
        os.makedirs(out, exist_ok=True)
        for i in range(count):
            stem = f"{tag}_{i:02d}"
            x, y = noise(N), noise(N)
            _write(os.path.join(out, f"{stem}_dry.wav"), x)
            _write(os.path.join(out, f"{stem}_wet.wav"), y)
            torch.save({"mod_sig": torch.from_numpy(rng.uniform(0, 1, size=(N // 100,)).astype(np.float32)),
                        "fx_params": {"rate_hz": float(rng.uniform(0.5, 3.0)), "shape": "cos", "tag": stem}},
                       os.path.join(out, f"{stem}.pt"))
    return {"dry": dry, "wet": wet, "pre_a": os.path.join(root, "pre_a"), "pre_b": os.path.join(root, "pre_b")}
