"""Where do the aten::add launches of one headline train step come from?  (GPU box; prints python stacks of aten::add / copy_)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from torch.profiler import profile, ProfilerActivity

sys.argv = ["bench.py", "--worker", "--config", "3", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-fp32-leg"]
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    try:
        bench.main()
    except SystemExit:
        pass
import collections
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::add", "aten::add_", "aten::copy_", "aten::uniform_", "aten::fill_", "aten::zero_"):
        st = [s for s in ev.stack if "mod_extraction_amd" in s or "bench.py" in s][:2]
        cnt[(ev.name, tuple(st))] += 1
for (name, st), n in cnt.most_common(30):
    print(n, name, " <- ".join(st))
