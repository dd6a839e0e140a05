"""Per-dispatch clock estimate from a rocprofv3 --pmc GRBM_GUI_ACTIVE (+ --kernel-trace) run:
python tools/pmc_clock.py <dir> [name filter]  ->  dispatch id, kernel, ms, GHz (GRBM_GUI_ACTIVE is summed over 8 XCDs)."""
import csv, glob, os, sys
d = sys.argv[1]; filt = sys.argv[2] if len(sys.argv) > 2 else ""
cc = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
rows = {}
for r in csv.DictReader(open(cc)):
    if filt and filt not in r["Kernel_Name"]: continue
    k = int(r["Dispatch_Id"])
    e = rows.setdefault(k, {"name": r["Kernel_Name"][:48], "ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
    e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for k in sorted(rows):
    e = rows[k]
    ghz = e.get("GRBM_GUI_ACTIVE", 0.0) / 8 / max(e["ns"], 1)
    extra = "  ".join(f"{n}={v:.4g}" for n, v in e.items() if n not in ("name", "ns", "GRBM_GUI_ACTIVE"))
    print(f"{k:5d} {e['name']:48s} {e['ns']/1e6:8.3f} ms  {ghz:5.2f} GHz  {extra}")
