"""Dev tool: per-iteration summary of a rocprofv3 kernel_stats.csv (usage: kernel_stats_summary.py <csv> <iterations> [rows])."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
it = float(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"kernel time per iteration {tot / 1e6 / it:.2f} ms, {sum(int(r['Calls']) for r in rows) / it:.0f} launches")
for r in rows[:top]:
    n = r["Name"]
    short = n[:70] + (".." + n[-30:] if len(n) > 100 else n[70:100])
    print(f"{int(r['TotalDurationNs']) / 1e6 / it:8.2f} ms/it {int(r['Calls']) / it:8.1f} calls/it  avg {float(r['AverageNs']) / 1e3:8.1f} us  {short}")
