"""Idle time between consecutive kernels of the TBPTT loop, from a rocprofv3 --kernel-trace csv:
    python tools/trace_gaps.py <dir with *_kernel_trace.csv>
Prints, for the stream that runs lstm_bwd_kernel, the kernel sequence of one optimizer step with the gap
before each kernel (start - previous end), and the totals over the whole trace."""
import csv
import glob
import os
import sys
from collections import defaultdict

path = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(path)))
key = "Stream_Id" if "Stream_Id" in rows[0] else ("Queue_Id" if "Queue_Id" in rows[0] else None)
by = defaultdict(list)
for r in rows:
    by[r[key] if key else "0"].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40]))
main = max(by, key=lambda k: sum(1 for x in by[k] if x[2].startswith("lstm_bwd")))
seq = sorted(by[main])
idx = [i for i, x in enumerate(seq) if x[2].startswith("lstm_bwd")]
print(f"stream/queue {main}: {len(seq)} kernels, {len(idx)} lstm_bwd launches")
if len(idx) > 12:
    a, b = idx[10], idx[11]
    for i in range(a, b + 1):
        s, e, n = seq[i]
        print(f"  gap {(s - seq[i - 1][1]) / 1e3:7.1f} us   run {(e - s) / 1e3:8.1f} us   {n}")
    lo, hi = idx[1], idx[-1]
    busy = sum(seq[i][1] - seq[i][0] for i in range(lo + 1, hi + 1))
    span = seq[hi][1] - seq[lo][1]
    print(f"between the first and last lstm_bwd: span {span / 1e6:.2f} ms, kernels {busy / 1e6:.2f} ms, idle {(span - busy) / 1e6:.2f} ms "
          f"= {(span - busy) / 1e3 / (len(idx) - 2):.1f} us per optimizer step")
