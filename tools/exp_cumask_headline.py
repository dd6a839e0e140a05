"""Experiment: headline step (config 3) with the side stream (batch render: phaser / flanger) confined to a few XCDs.
A CU-masked side stream next to torch's DEFAULT main stream serialised the two (75.9 -> 86.5 ms); here BOTH streams are
created with hipExtStreamCreateWithCUMask (the main one with every CU enabled, or with the complement).
    python tools/exp_cumask_headline.py [2]        # 2: config 2 (all phaser, bs 64) instead of the headline
"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
bench.torch = torch          # (bench.py imports torch lazily in main)
from mod_extraction_amd import trainer as tr

hip = ctypes.CDLL("libamdhip64.so")
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
torch.zeros(1, device=dev)
F = 0xFFFFFFFF


def masked_stream(words):
    st = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), len(words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value, device=dev)


CFG2 = len(sys.argv) > 1 and sys.argv[1] == "2"          # config 2: all phaser, 64 clips


def run(main_words, side_words, steps=12):
    module, opt, batcher = bench.build_lfo_job(dev, 0, 64 if CFG2 else 256, ("phaser",) if CFG2 else ("flanger", "chorus", "phaser"))
    runner = tr.Trainer(log_fn=None)
    if side_words is not None:
        batcher.use_side_stream(masked_stream(side_words))
    main = masked_stream(main_words) if main_words is not None else torch.cuda.current_stream()
    main.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(main):
        for _ in range(3):
            runner.train_step(module, opt, batcher.next_batch())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            runner.train_step(module, opt, batcher.next_batch())
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


layouts = {
    "default streams": (None, None),
    "main all CUs (masked stream) / side default": ([F] * 8, None),
    "main all CUs / side XCD 7": ([F] * 8, [0] * 7 + [F]),
    "main all CUs / side XCDs 6-7": ([F] * 8, [0] * 6 + [F] * 2),
    "main XCDs 0-6 / side XCD 7": ([F] * 7 + [0], [0] * 7 + [F]),
    "main all CUs / side 8 CUs of every XCD": ([F] * 8, [0xFF] * 8),
}
for name, (m, s) in layouts.items():
    try:
        print(f"{name:50s} {run(m, s):7.2f} ms per step", flush=True)
    except Exception as e:
        print(name, "FAILED", repr(e)[:300], flush=True)
