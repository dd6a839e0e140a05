"""Summarise a rocprofv3 --pmc counter_collection.csv (+ kernel_trace.csv) per kernel name.
    python tools/pmc_summary.py <dir with *_counter_collection.csv> [substring filter]
"""
import csv, glob, os, sys, collections
d = sys.argv[1]; filt = sys.argv[2] if len(sys.argv) > 2 else ""
cc = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
kt = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)
dur = {}
if kt:
    for r in csv.DictReader(open(kt[0])):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set); t = collections.defaultdict(float)
for r in csv.DictReader(open(cc)):
    name = r["Kernel_Name"]
    if filt and filt not in name: continue
    acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in n[name]:
        n[name].add(r["Dispatch_Id"]); t[name] += dur.get(r["Dispatch_Id"], 0)
for name in sorted(acc, key=lambda k: -t[k]):
    c = acc[name]; k = len(n[name])
    print(f"{name[:60]:60s} x{k:3d} {t[name]/1e6/max(k,1):9.3f} ms/launch  " + "  ".join(f"{cn}={cv/k:.4g}" for cn, cv in sorted(c.items())))
