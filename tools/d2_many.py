#!/usr/bin/env python3
"""Many-seed return-curve runs for the config-3 parity question: a pool of concurrent single-seed
`return_curve_parity.py` processes on ONE GPU (1024-env runs are launch-latency-bound, so several processes share the
device well) and, beside them, CPU-oracle seeds on the host's cores.  Every job writes its own small JSON (only the
Train/* curves + the four episode terms the merge reads, rounded), so a call that is cut off keeps what finished.

  python tools/d2_many.py --out gpurun_out/d2r4 --arms fast:1-48 eager:1-48 --workers 12 [--cpu_seeds 11-18 --cpu_threads 16]

arms: fast (default product path), eager (QA_PARITY_EAGER=1: no recorded update steps, no stream overlap), seq (recorded, QA_OVERLAP_UPDATES=0),
      nodisc (recorded PPO only), noac (recorded discriminator only), eagerupd (QA_PARITY_EAGER_UPDATE=1: eager update, overlap flag untouched)
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARM_ENV = {
    "fast": {},
    "eager": {"QA_PARITY_EAGER": "1"},
    "seq": {"QA_OVERLAP_UPDATES": "0"},
    "nodisc": {"QA_PARITY_NO_DISC_GRAPH": "1"},
    "noac": {"QA_PARITY_NO_AC_GRAPH": "1"},
    "eagerupd": {"QA_PARITY_EAGER_UPDATE": "1"},
    "hybrid": {"QA_D2_SIDE": "hybrid"},        # r5: oracle physics on the host cores + the GPU learner (tools/hybrid_backend.py); OMP threads per job = --hybrid_threads
}
KEEP = ("Train/mean_reward", "Train/mean_reward_t", "Train/mean_reward_i", "Train/mean_episode_length", "Episode/rew_tracking_lin_vel",
        "Episode/rew_tracking_ang_vel", "Episode/rew_collision", "Episode/rew_dof_error", "Episode/rew_torques")


def seeds_of(spec):
    out = []
    for part in spec.split(","):
        if "-" in part:
            a, b = part.split("-"); out += list(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return out


def trim(path):
    d = json.load(open(path))
    for r in d["rows"]:
        r["curves"] = {k: [float(f"{x:.6g}") for x in v] for k, v in r["curves"].items() if k in KEEP}
    json.dump(d, open(path, "w"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--arms", nargs="*", default=[])
    ap.add_argument("--workers", type=int, default=12)
    ap.add_argument("--num_envs", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=1000)
    ap.add_argument("--cpu_seeds", default="")
    ap.add_argument("--cpu_threads", type=int, default=16)
    ap.add_argument("--job_timeout", type=int, default=1500)
    ap.add_argument("--budget_s", type=int, default=10 ** 9, help="stop STARTING jobs after this many seconds")
    ap.add_argument("--plain", action="store_true", help="config 2 (no --amp)")
    ap.add_argument("--hybrid_threads", type=int, default=24, help="OpenMP threads of one hybrid-arm job (oracle physics)")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    jobs = []
    arms = [(s.split(":")[0], seeds_of(s.split(":")[1])) for s in a.arms]
    # interleave the arms so that a cut-off call leaves balanced samples
    for k in range(max((len(s) for _, s in arms), default=0)):
        for arm, ss in arms:
            if k < len(ss):
                jobs.append((arm, ss[k]))
    amp = [] if a.plain else ["--amp"]
    cpu_procs = []
    for s in (seeds_of(a.cpu_seeds) if a.cpu_seeds else []):
        out = os.path.join(a.out, f"cpu_s{s}.json")
        if os.path.exists(out):
            continue
        env = dict(os.environ, OMP_NUM_THREADS=str(a.cpu_threads), QA_CPU_THREADS=str(a.cpu_threads), QA_PARITY_LOG_ROOT=os.path.join(a.out, f"cpu_s{s}_log"),
                   HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
        cmd = [sys.executable, os.path.join(ROOT, "tools", "return_curve_parity.py"), "--side", "cpu", "--num_envs", str(a.num_envs), "--iters", str(a.iters),
               "--seeds", str(s), "--out", out] + amp
        cpu_procs.append((s, out, subprocess.Popen(cmd, env=env, stdout=open(out.replace(".json", ".log"), "w"), stderr=subprocess.STDOUT, stdin=subprocess.DEVNULL)))
    t0 = time.time()
    running, done = [], 0
    while jobs or running:
        for r in list(running):
            arm, s, out, p, ts = r
            rc = p.poll()
            if rc is None and time.time() - ts > a.job_timeout:
                p.kill(); rc = -9
            if rc is not None:
                running.remove(r); done += 1
                if rc == 0 and os.path.exists(out):
                    trim(out)
                print(f"[{time.time() - t0:7.0f}s] {arm} seed {s}: rc {rc} in {time.time() - ts:.0f}s ({done} done, {len(jobs)} queued)", flush=True)
        while jobs and len(running) < a.workers and time.time() - t0 < a.budget_s:
            arm, s = jobs.pop(0)
            out = os.path.join(a.out, f"{arm}_s{s}.json")
            if os.path.exists(out):
                continue
            env = dict(os.environ, **ARM_ENV[arm])
            side = env.pop("QA_D2_SIDE", "gpu")
            if side == "hybrid":
                env["OMP_NUM_THREADS"] = str(a.hybrid_threads)
            cmd = [sys.executable, os.path.join(ROOT, "tools", "return_curve_parity.py"), "--side", side, "--num_envs", str(a.num_envs), "--iters", str(a.iters),
                   "--seeds", str(s), "--out", out] + amp
            p = subprocess.Popen(cmd, env=env, stdout=open(out.replace(".json", ".log"), "w"), stderr=subprocess.STDOUT, stdin=subprocess.DEVNULL)
            running.append((arm, s, out, p, time.time()))
        if jobs and not running and time.time() - t0 >= a.budget_s:
            print(f"budget reached with {len(jobs)} jobs not started", flush=True)
            break
        time.sleep(1.0)
    for s, out, p in cpu_procs:
        left = max(1.0, a.budget_s + 600 - (time.time() - t0)) if a.budget_s < 10 ** 8 else None
        try:
            rc = p.wait(timeout=left)
        except subprocess.TimeoutExpired:
            p.kill(); rc = -9
        if rc == 0 and os.path.exists(out):
            trim(out)
        for dp, _, fs in os.walk(os.path.join(a.out, f"cpu_s{s}_log")):      # keep only the scalar log of a run that was cut off (checkpoints are 12 MB each)
            for f in fs:
                if f != "scalars.jsonl" or (rc == 0 and os.path.exists(out)):
                    os.remove(os.path.join(dp, f))
        print(f"[{time.time() - t0:7.0f}s] cpu seed {s}: rc {rc}", flush=True)


if __name__ == "__main__":
    main()
