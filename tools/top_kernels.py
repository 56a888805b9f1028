"""Print the top-N rows of a rocprofv3 kernel_stats.csv with shortened names (dev tool)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms over {sum(int(r['Calls']) for r in rows)} launches")
for r in rows[:n]:
    name = r["Name"]
    if name.startswith("Cijk"):
        name = "GEMM " + name[:40] + ".." + (name.split("_MT")[1][:12] if "_MT" in name else "")
    print(f"{float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['Percentage']):6.2f}% calls={int(r['Calls']):7d} avg={float(r['AverageNs'])/1e3:9.1f} us  {name[:110]}")
