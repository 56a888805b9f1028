"""Dev tool: qa_mlp_forward (one launch) vs. the GEMM path for the rollout's policy inference, at N envs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quadrupedal_agility_amd.rsl_rl.algorithms.fused import PolicyChain
from tests.test_policy_chain import modules, torch_reference

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
REPS = 200
ac, est, n_obs = modules(seed=1)
ac, est = ac.cuda(), est.cuda()
obs = torch.randn(N, n_obs, device="cuda")
chain = PolicyChain.describe(ac, est, True)
chain.pack()


def timed(fn):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(REPS):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS * 1e3


t_chain = timed(lambda: chain.forward(obs))
t_pack = timed(chain.pack)
with torch.inference_mode():
    t_torch = timed(lambda: torch_reference(ac, est, obs, True))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        torch_reference(ac, est, obs, True)
    t_graph = timed(g.replay)
hchain = PolicyChain.describe(ac, est, False, hist_encoding=True, with_critic=False)      # what play.py / the exported policy evaluates
with torch.inference_mode():
    hchain.pack()
    t_hist = timed(lambda: hchain.forward(obs))
    t_hist_torch = timed(lambda: ac.act_inference(obs, hist_encoding=True))
    gh = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gh):
        ac.act_inference(obs, hist_encoding=True)
    t_hist_graph = timed(gh.replay)
print(f"N={N}: act_inference(history encoder): qa_mlp_forward {t_hist:.1f} us, torch eager {t_hist_torch:.1f} us, torch in a hipGraph {t_hist_graph:.1f} us")
flops = 1488128 * N
print(f"N={N}: qa_mlp_forward {t_chain:.1f} us ({flops / t_chain / 1e6:.1f} TFLOP/s, {flops / t_chain / 1e6 / 157.3 * 100:.1f}% of fp32 MFMA peak), "
      f"qa_mlp_pack {t_pack:.1f} us, torch eager {t_torch:.1f} us, torch in a hipGraph {t_graph:.1f} us")
