"""Workload + summary for the SQ counter pass of the policy kernel:
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU \\
            --kernel-trace --output-format csv -d /tmp/pmc_p -- python tools/pmc_policy.py run 4096
  python tools/pmc_policy.py summarize /tmp/pmc_p
(SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over wavefronts; SQ_VALU_MFMA_BUSY_CYCLES counts cycles.)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "run":
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
    for f in glob.glob(sys.argv[2] + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "qa_mlp_forward" in r.get("Kernel_Name", ""):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {c: sum(v[len(v) // 4:]) / len(v[len(v) // 4:]) for c, v in acc.items()}
    for c, v in sorted(m.items()):
        print(f"{c:28s} {v:16.0f}")
    if "SQ_WAVE_CYCLES" in m:
        wc = m["SQ_WAVE_CYCLES"]
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if c in m:
                print(f"{c} / SQ_WAVE_CYCLES = {m[c] / wc:.3f}")
