"""Does the headline train step write its parameter gradients in place?  (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from mod_extraction_amd import models, trainer as tr

orig = models._direct_grad_views


def probe(params):
    out = orig(params)
    why = []
    base = None
    for i, p in enumerate(params):
        g = p.grad if p.is_leaf else None
        if not (p.is_leaf and p.requires_grad):
            why.append(f"param {i}: leaf={p.is_leaf} requires_grad={p.requires_grad}")
        elif g is None:
            why.append(f"param {i}: grad None")
        elif g._base is None:
            why.append(f"param {i}: grad is not a view")
        elif not g.is_contiguous():
            why.append(f"param {i}: grad not contiguous")
        else:
            base = g._base
    print("direct views:", None if out is None else len(out), "| fresh flag on base:", None if base is None else getattr(base, "_modex_fresh", "absent"),
          "|", "; ".join(why[:4]), flush=True)
    return out


models._direct_grad_views = probe
bench.torch = torch
device = torch.device("cuda", 0)
module, opt, batcher = bench.build_lfo_job(device, 0, 16, bench.CONFIGS[3]["kinds"], overlap=True)
runner = tr.Trainer(log_fn=None)
for _ in range(2):
    runner.train_step(module, opt, batcher.next_batch())
torch.cuda.synchronize()
print("params in optimizer:", len(list(opt.params)) if hasattr(opt, "params") else "?", "| module params:", len(list(module.parameters())))
