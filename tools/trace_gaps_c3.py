"""Idle time between consecutive kernels of the headline train step, from a rocprofv3 --kernel-trace csv:
    python tools/trace_gaps_c3.py <dir with *_kernel_trace.csv>
For the stream that runs the forward convolutions: per step (delimited by adamw_kernel) the busy time, the idle time and the
largest gaps with the kernels on either side."""
import csv, glob, os, sys
from collections import defaultdict

path = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(path)))
key = "Stream_Id" if "Stream_Id" in rows[0] else ("Queue_Id" if "Queue_Id" in rows[0] else None)
by = defaultdict(list)
for r in rows:
    by[r[key] if key else "0"].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:44]))
main = max(by, key=lambda k: sum(1 for x in by[k] if "dma16" in x[2]))
seq = sorted(by[main])
idx = [i for i, x in enumerate(seq) if x[2].startswith("adamw")]
print(f"stream/queue {main}: {len(seq)} kernels, {len(idx)} optimizer steps")
for a, b in zip(idx[1:-1], idx[2:]):
    span = seq[b][1] - seq[a][1]
    busy = sum(seq[i][1] - seq[i][0] for i in range(a + 1, b + 1))
    gaps = sorted(((seq[i][0] - seq[i - 1][1], seq[i - 1][2], seq[i][2]) for i in range(a + 1, b + 1)), reverse=True)
    print(f"step: span {span / 1e6:.2f} ms, kernels {busy / 1e6:.2f} ms, idle {(span - busy) / 1e6:.2f} ms over {b - a} launches; largest gaps:")
    for g, p, n in gaps[:8]:
        print(f"     {g / 1e3:8.1f} us  between {p}  and  {n}")
