"""Experiment: the headline train step on ONE fixed batch (nothing rendered, nothing on the side stream) against the bench's
step with the next batch rendered concurrently -- what the side-stream effect kernels cost the main stream.
    python tools/exp_cnn_step_only.py
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
bench.torch = torch          # (bench.py imports torch lazily in main)
from mod_extraction_amd import trainer as tr

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)


def run(fixed, steps=30, kinds=("flanger", "chorus", "phaser")):
    module, opt, batcher = bench.build_lfo_job(dev, 0, 256, kinds, overlap=not fixed)
    runner = tr.Trainer(log_fn=None)
    batch = batcher.next_batch()
    if fixed:
        batch = tuple(t.clone() if isinstance(t, torch.Tensor) else t for t in batch)
    nxt = (lambda: batch) if fixed else batcher.next_batch
    for _ in range(3):
        runner.train_step(module, opt, nxt())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        runner.train_step(module, opt, nxt())
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


print(f"fixed batch, no rendering      {run(True):7.2f} ms per step")
print(f"next batch rendered alongside  {run(False):7.2f} ms per step")
print(f"  ... 256 flanger clips only   {run(False, kinds=('flanger',)):7.2f} ms per step")
print(f"  ... 256 phaser clips only    {run(False, kinds=('phaser',)):7.2f} ms per step")
print(f"  ... 256 unprocessed clips    {run(False, kinds=('dry',)):7.2f} ms per step (noise + LFO kernels only)")
