"""Audit: which torch reductions run inside the code we record into hipGraphs, and do they survive replay?

Finding behind this tool (profiles/r2_hipgraph_stale_reductions.md): on ROCm 7.2 / torch 2.10 a torch reduction that takes the two-stage
"global reduce" path (tall-skinny `sum(0)`, `mean()` / `sum()` / `max()` over a few hundred thousand elements) returns the result of its FIRST
execution when replayed from a hipGraph on new data -- the second stage does not re-run.  Our own kernels reduce in a fixed order without
semaphores and are not affected.

  1. run one training iteration of each tree EAGERLY under a TorchDispatchMode and collect every reduction (op, shape, dim) that the code which
     is normally recorded executes (forward and backward);
  2. replay each of them from a hipGraph on fresh data and compare with the eager result.

Necessary, not sufficient: the discriminator's (921, 512) bias-gradient `sum(0)` passes step 2 stand-alone and still left its output unwritten
inside the recorded discriminator step.  The rule the code follows is therefore "no torch reduction over the batch dimension in a gradient path
of a recorded step"; this tool is the list to read against that rule.

    python tools/graph_reduction_audit.py [--envs 1024] [--json out.json]
"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode

REDUCTIONS = {"sum", "mean", "amax", "amin", "max", "min", "norm", "linalg_vector_norm", "var", "std", "var_mean", "std_mean", "prod", "logsumexp",
              "argmax", "argmin", "all", "any", "nansum", "count_nonzero", "_softmax", "_log_softmax", "_softmax_backward_data",
              "_log_softmax_backward_data", "cumsum", "median", "dot", "vdot", "mse_loss", "mse_loss_backward", "native_layer_norm",
              "native_batch_norm", "native_batch_norm_backward", "cudnn_batch_norm", "miopen_batch_norm", "miopen_batch_norm_backward",
              "native_layer_norm_backward", "binary_cross_entropy_with_logits", "nll_loss_forward", "index_add", "index_put", "scatter_add",
              "embedding_dense_backward", "unfold_backward", "mv", "addmv"}


class Collect(TorchDispatchMode):
    def __init__(self, where, seen):
        super().__init__(); self.where, self.seen = where, seen

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        name = func.overloadpacket.__name__
        if name in REDUCTIONS and args and torch.is_tensor(args[0]) and args[0].is_cuda:
            shapes = tuple(tuple(a.shape) if torch.is_tensor(a) else (a if isinstance(a, (int, float, bool, list, tuple, type(None))) else str(type(a)))
                           for a in args)
            key = (str(func), json.dumps(shapes, default=str), json.dumps({k: str(v) for k, v in kwargs.items()}))
            self.seen.setdefault(key, set()).add(self.where)
        return func(*args, **kwargs)


def replay_check(func_name, shapes, kwargs):
    """run aten op `func_name` on random tensors of `shapes` eagerly and from a graph, on data that changes between replays"""
    ns, op = func_name.split(".", 1)
    packet, overload = op.rsplit(".", 1)
    func = getattr(getattr(getattr(torch.ops, ns), packet), overload)
    shapes = json.loads(shapes)
    schema = func._schema                                   # tensor shapes and int lists both come back as lists: the schema says which is which
    args = []
    for a, s in zip(schema.arguments, shapes):
        if "Tensor" in str(a.type) and s is not None:
            t = torch.randn(*s, device="cuda")
            if "index" in a.name or "indices" in a.name:
                t = torch.randint(0, 2, tuple(s), device="cuda")
            args.append(t)
        else:
            args.append(s)
    try:
        ref_fn = lambda: func(*args)
        ref_fn()
        s_ = torch.cuda.Stream(); s_.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s_): ref_fn()
        torch.cuda.current_stream().wait_stream(s_)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g): out = ref_fn()
        worst = 0.0
        for r in range(3):
            for t in args:
                if torch.is_tensor(t) and t.is_floating_point(): t.copy_(torch.randn_like(t))
            want = ref_fn()
            g.replay(); torch.cuda.synchronize()
            outs = out if isinstance(out, (tuple, list)) else [out]
            wants = want if isinstance(want, (tuple, list)) else [want]
            for a, b in zip(outs, wants):
                if torch.is_tensor(a) and a.numel():
                    d = (a.double() - b.double()).abs().max().item(); sc = b.double().abs().max().item() + 1e-30
                    worst = max(worst, d / sc)
        return worst
    except Exception as e:                                  # an op we cannot re-create stand-alone (integer inputs, ...)
        return f"not re-run: {type(e).__name__}: {str(e)[:80]}"


def collect_bbc(envs, amp, seen):
    from tests.test_gpu_train import _make
    from quadrupedal_agility_amd.legged_gym.envs import task_registry
    os.environ["QA_ROLLOUT_GRAPH"] = "0"
    torch.manual_seed(0)
    env, args, tcfg = _make(envs, amp)
    runner, _ = task_registry.make_alg_runner(env, name="go2_locomotion", args=args, train_cfg=tcfg, log_root=None)
    runner.alg.use_update_graph = False; runner.alg.overlap_updates = False
    with Collect("bbc-amp" if amp else "bbc", seen):
        runner.learn(2, init_at_random_ep_len=True)           # iteration 0 is a DAgger iteration


def collect_tsc(envs, vision, seen):
    from quadrupedal_agility_amd.legged_gym.utils.cfg_to_c import class_to_dict
    from quadrupedal_agility_amd.tsc.legged_gym.envs.base import legged_robot as lr
    from quadrupedal_agility_amd.tsc.legged_gym.envs.go2.go2_agility_config import Go2AgilityCfg, Go2AgilityCfgPPO
    from quadrupedal_agility_amd.tsc.rsl_rl.runners import OnPolicyRunner
    os.environ["QA_TSC_ROLLOUT_GRAPH"] = "0"; os.environ["QA_TSC_UPDATE_GRAPH"] = "0"
    cfg = Go2AgilityCfg(); cfg.env.num_envs, cfg.seed, cfg.course_seed = envs, 1, 1
    cfg.depth.use_camera = vision
    tcfg = class_to_dict(Go2AgilityCfgPPO()); tcfg["depth_encoder"]["if_depth"] = vision
    torch.manual_seed(1)
    env = lr.LeggedRobot(cfg, sim_device="cuda:0")
    runner = OnPolicyRunner(env, tcfg, log_dir=None, device="cuda:0")
    with Collect("tsc-student" if vision else "tsc-teacher", seen):
        (runner.learn_vision if vision else runner.learn)(2, init_at_random_ep_len=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=1024)
    ap.add_argument("--json", default=None)
    ap.add_argument("--trees", default="bbc,bbc-amp,tsc")
    a = ap.parse_args()
    seen = {}
    for tree in a.trees.split(","):
        if tree == "bbc": collect_bbc(a.envs, False, seen)
        elif tree == "bbc-amp": collect_bbc(a.envs, True, seen)
        elif tree == "tsc": collect_tsc(a.envs, False, seen)
        elif tree == "tsc-student": collect_tsc(min(a.envs, 256), True, seen)
    rows = []
    for (fn, shapes, kw), where in sorted(seen.items()):
        err = replay_check(fn, shapes, kw)
        stale = (not isinstance(err, str)) and err > 1e-5
        rows.append(dict(op=fn, args=shapes, where=sorted(where), replay_rel_err=err, stale=stale))
        if stale or isinstance(err, str):
            print("STALE " if stale else "??    ", fn, shapes, sorted(where), err)
    print(f"{len(rows)} distinct reductions, {sum(r['stale'] for r in rows)} stale under replay, {sum(isinstance(r['replay_rel_err'], str) for r in rows)} not re-run")
    if a.json:
        json.dump(rows, open(a.json, "w"), indent=1)
