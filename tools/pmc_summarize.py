"""Summarise rocprofv3 counter_collection.csv files: mean counter value per kernel name (dev tool)."""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        short = "qa_env_step_kernel" if "qa_env_step" in name else ("copy(calibration)" if ("copy" in name.lower() or "Copy" in name) else None)
        if short:
            acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    for c, v in d.items():
        v2 = v[len(v) // 4:]          # skip warm-up dispatches
        print(f"{k:24s} {c:12s} n={len(v2):4d} mean={sum(v2)/len(v2):14.1f} min={min(v2):14.1f} max={max(v2):14.1f}")
