"""Dev tool: where does the GPU idle?  Reads a rocprofv3 --kernel-trace CSV (kernel_trace.csv), sorts dispatches by start time
and aggregates the idle gaps by (kernel before, kernel after).  usage: gap_report.py <kernel_trace.csv> [skip_fraction]"""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6          # ignore the warm-up part of the run
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
ev = ev[int(len(ev) * skip):]
busy = sum(e - s for s, e, _ in ev)
span = ev[-1][1] - ev[0][0]
gaps = defaultdict(lambda: [0, 0])
end = ev[0][1]
prev = ev[0][2]
for s, e, n in ev[1:]:
    if s > end:
        g = gaps[(prev[:60], n[:60])]
        g[0] += s - end; g[1] += 1
    if e > end:
        end, prev = e, n
idle = sum(g[0] for g in gaps.values())
print(f"window {span / 1e6:.1f} ms: busy {busy / 1e6:.1f} ms, idle {idle / 1e6:.1f} ms ({100 * idle / span:.1f}%), {len(ev)} dispatches")
for (a, b), (t, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"{t / 1e6:8.2f} ms {c:6d}x avg {t / c / 1e3:8.1f} us   {a}  ->  {b}")
