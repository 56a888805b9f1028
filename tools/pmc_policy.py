"""Workload + summary for the SQ counter pass of the policy kernel:
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU \\
            --kernel-trace --output-format csv -d /tmp/pmc_p -- python tools/pmc_policy.py run 4096
  python tools/pmc_policy.py summarize /tmp/pmc_p
(SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over wavefronts; SQ_VALU_MFMA_BUSY_CYCLES counts cycles.)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "train":
    # r6: the PPO TRAINING step's launches at the 512-env share (3,072 rows): forward chain, input-gradient chain, the batched weight-gradient products
    import torch
    from quadrupedal_agility_amd.rsl_rl.algorithms.train_chain import PpoTrainChain
    from tests.test_policy_chain import modules
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 3072
    ac, est, n_obs = modules(seed=1)
    ac, est = ac.cuda(), est.cuda()
    chain = PpoTrainChain.describe(ac, est, n)
    assert chain is not None
    obs = torch.randn(n, 672, device="cuda")[:, :n_obs]
    d = chain.dims
    g = lambda w: torch.randn(n, w, device="cuda") / n
    g_est, dmu, dvalue, g_priv = g(d["n_exp"]), g(d["n_act"]), g(1), g(d["n_lat"])
    with torch.no_grad():
        for _ in range(30):
            chain.pack()
            chain.forward(obs)
            chain.backward(g_est, dmu, dvalue, g_priv, defer=False)
    torch.cuda.synchronize()
    print("done")
elif sys.argv[1] == "run":
    import torch
    from quadrupedal_agility_amd.rsl_rl.algorithms.fused import PolicyChain
    from tests.test_policy_chain import modules
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    ac, est, n_obs = modules(seed=1)
    ac, est = ac.cuda(), est.cuda()
    chain = PolicyChain.describe(ac, est, True)
    obs = torch.randn(n, n_obs, device="cuda")
    with torch.inference_mode():
        chain.pack()
        for _ in range(40):
            chain.forward(obs)
    torch.cuda.synchronize()
    print("done")
else:
    import csv, glob, collections
    acc = collections.defaultdict(list)
    split = len(sys.argv) > 3 and sys.argv[3] == "train"        # the training step: forward / input-gradient chain launches alternate; the product groups by name
    groups = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(sys.argv[2] + "/**/*counter_collection.csv", recursive=True):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r.get("Dispatch_Id", 0)))
        seen = collections.defaultdict(int)
        for r in rows:
            name = r.get("Kernel_Name", "")
            if "qa_mlp_forward" in name:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                if split:
                    k = seen[r["Counter_Name"]]; seen[r["Counter_Name"]] += 1
                    groups["forward chain" if k % 2 == 0 else "input-gradient chain"][r["Counter_Name"]].append(float(r["Counter_Value"]))
            elif split and "qa_wgrad_group_kernel" in name:
                groups[name.split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for gname, dd in groups.items():
        mm = {c: sum(v[len(v) // 4:]) / len(v[len(v) // 4:]) for c, v in dd.items()}
        print(f"--- {gname}")
        for c, v in sorted(mm.items()):
            print(f"{c:28s} {v:16.0f}")
        if "SQ_WAVE_CYCLES" in mm:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
                if c in mm:
                    print(f"{c} / SQ_WAVE_CYCLES = {mm[c] / mm['SQ_WAVE_CYCLES']:.3f}")
    if split:
        sys.exit(0)
    m = {c: sum(v[len(v) // 4:]) / len(v[len(v) // 4:]) for c, v in acc.items()}
    for c, v in sorted(m.items()):
        print(f"{c:28s} {v:16.0f}")
    if "SQ_WAVE_CYCLES" in m:
        wc = m["SQ_WAVE_CYCLES"]
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if c in m:
                print(f"{c} / SQ_WAVE_CYCLES = {m[c] / wc:.3f}")
