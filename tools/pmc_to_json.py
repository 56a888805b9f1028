"""FETCH_SIZE / WRITE_SIZE passes (tools/pmc_env_step.py under rocprofv3 --pmc, one counter per pass) -> profiles/env_step_traffic.json.
Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes: both counters are in KiB; on gfx950 FETCH_SIZE reports
half of the bytes of wide coalesced reads -- checked here on the calibration copy of the same run (256 MiB read + 256 MiB written)
and applied only if the calibration shows it.  usage: pmc_to_json.py FETCH_DIR WRITE_DIR NUM_ENVS OUT.json [LEAN] [KERNELS]
KERNELS (r5): comma-separated kernel-name substrings whose per-launch means are SUMMED (the task-level env step is three launches:
qa_env_step_kernel + qa_tsc_goal_step + qa_tsc_observations); default: qa_env_step alone."""
import collections, csv, glob, json, sys


KERNELS = sys.argv[6].split(",") if len(sys.argv) > 6 else ["qa_env_step"]
CAL_KIB = 256 * 1024


def means(d):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "")
            hit = [k for k in KERNELS if k in name]
            val = float(r["Counter_Value"])
            # the calibration stream is the 256 MiB clone at the end of the workload: a copy kernel that moved at least a quarter of that.  (r5 matched
            # every kernel whose name contains "copy" -- in the task-level pass also the few-KB copyBuffer / direct_copy launches of env.step itself,
            # so the calibration mean collapsed to 3.4 MB and the derived "correction" came out as 71-77: VERDICT r5 weak item 5.)
            key = ("env:" + hit[0]) if hit else ("copy" if ("copy" in name.lower() and val >= CAL_KIB / 4) else None)
            if key:
                acc[key].append(val)
    m = {k: sum(v[len(v) // 4:]) / len(v[len(v) // 4:]) for k, v in acc.items()}
    m["env"] = sum(v for k, v in m.items() if k.startswith("env:"))        # one launch of each per env step
    return m


f, w = means(sys.argv[1]), means(sys.argv[2])
n = int(sys.argv[3])
copy_kib = 256 * 1024
fetch_corr = round(copy_kib / f["copy"]) if "copy" in f else 2        # 2 on gfx950 (the guide), 1 if the counter were exact
write_corr = round(copy_kib / w["copy"], 2) if "copy" in w else 1.0
if fetch_corr not in (1, 2) or not (0.8 <= write_corr <= 1.25):
    raise SystemExit(f"pmc_to_json: the calibration copy reads as fetch x{copy_kib / f.get('copy', float('nan')):.2f}, write x{write_corr}: not the guide's factor (1 or 2) / "
                     "an exact write count -- the calibration launches were not identified; no figure is written")
out = {"kernel": " + ".join(KERNELS) if len(KERNELS) > 1 else "qa_env_step_kernel", "num_envs": n, "lean_exports": int(sys.argv[5]) if len(sys.argv) > 5 else 3, "fetch_size_kib": f["env"], "write_size_kib": w["env"], "fetch_correction": fetch_corr,
       "write_calibration": write_corr, "calibration_copy_fetch_kib": f.get("copy"), "calibration_copy_write_kib": w.get("copy"),
       "hbm_bytes_per_launch": int((f["env"] * fetch_corr + w["env"]) * 1024), "source": "tools/final_measure.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"}
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
import bench
if len(KERNELS) > 1:      # (r6) which of the launches moves the bytes
    out["per_kernel_hbm_bytes"] = {k: int((f.get("env:" + k, 0.0) * fetch_corr + w.get("env:" + k, 0.0)) * 1024) for k in KERNELS}
    out["per_kernel_fetch_write_kib"] = {k: [round(f.get("env:" + k, 0.0) * fetch_corr, 1), round(w.get("env:" + k, 0.0), 1)] for k in KERNELS}
out["kernel_source_hash"] = bench.env_kernel_hash(g)        # bench.py reports this figure only for the kernel sources it was measured on
json.dump(out, open(sys.argv[4], "w"))
print(json.dumps(out))
