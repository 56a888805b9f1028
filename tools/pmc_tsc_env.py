"""Workload for the counter passes of the task-level physics kernel (qa_env_step_kernel<false,4,1>, the course as collision terrain with
articulated obstacles and self-collision): N envs after a short random-action warm-up, 40 direct launches, plus the 256 MiB calibration copy.
  rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d DIR -- python tools/pmc_tsc_env.py 8192
  python tools/pmc_tsc_env.py summarize DIR [DIR2 ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] != "summarize":
    import torch
    from quadrupedal_agility_amd.tsc.legged_gym.envs.base import legged_robot as lr
    from quadrupedal_agility_amd.tsc.legged_gym.envs.go2.go2_agility_config import Go2AgilityCfg
    n = int(sys.argv[1])
    cfg = Go2AgilityCfg()
    cfg.env.num_envs, cfg.seed, cfg.course_seed = n, 1, 1
    d = cfg.domain_rand
    d.randomize_base_mass = d.randomize_base_com = d.push_robots = True
    cfg.obstacle.randomize_start = True
    env = lr.LeggedRobot(cfg, sim_device="cuda:0")
    g = torch.Generator(device="cuda").manual_seed(0)
    for _ in range(30):                                    # robots on and between the obstacles, some fallen
        env.sim.physics_step(torch.randn(n, 12, device="cuda", generator=g) * 0.5, 1)
    act = torch.randn(n, 12, device="cuda", generator=g) * 0.3
    if len(sys.argv) > 2 and sys.argv[2] == "full":       # (r5) the whole task-level env step: physics + goal step + reset + observations (bench.py --tsc's roofline kernels)
        hist = torch.zeros(n, 8, 19, device="cuda")
        for _ in range(40):
            env.step(act, hist)
    else:
        for _ in range(40):
            env.sim.physics_step(act, 1)
    x = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device="cuda").normal_()
    for _ in range(5):
        y = x.clone()
    torch.cuda.synchronize()
    print("done")
else:
    import collections, csv, glob
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sys.argv[2:]:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                name = r.get("Kernel_Name", "")
                key = name.split("(")[0].replace("void ", "")[:40] if "qa_env_step" in name else ("copy (256 MiB calibration)" if "copy" in name.lower() else None)
                if key:
                    acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, dd in acc.items():
        for c, v in sorted(dd.items()):
            v2 = v[len(v) // 4:]
            print(f"{k:42s} {c:28s} n={len(v2):3d} mean={sum(v2) / len(v2):16.1f}")
