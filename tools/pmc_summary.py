"""Summarise rocprofv3 --pmc CSV output (one row per dispatch and counter) for the kernels whose name contains a substring.
Usage: python tools/pmc_summary.py <dir-with-*counter_collection.csv> <kernel substring> [skip-first-n]
Prints JSON: per counter the mean value per dispatch, plus the mean kernel duration."""
import csv, glob, json, os, sys
d, sub = sys.argv[1], sys.argv[2]
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
acc, dur, seen = {}, [], {}
for f in files:
    for r in csv.DictReader(open(f)):
        if sub not in r["Kernel_Name"]:
            continue
        key = r["Dispatch_Id"]
        c = r["Counter_Name"]
        n = seen.setdefault((c,), {})
        n[key] = float(r["Counter_Value"])
        if "Start_Timestamp" in r and r.get("End_Timestamp"):
            seen.setdefault(("dur",), {})[key] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
out = {}
for (c,), vals in seen.items():
    ks = sorted(vals, key=lambda x: int(x))[skip:]
    if ks:
        out[c if c != "dur" else "avg_duration_us"] = sum(vals[k] for k in ks) / len(ks)
        out.setdefault("dispatches", len(ks))
print(json.dumps(out, indent=1))
