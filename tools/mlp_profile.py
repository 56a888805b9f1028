"""Dev tool: cycles per op of qa_mlp_forward (workgroup 0), from s_memtime stamps.  Builds a -DQA_MLP_PROF copy of
qa_policy.hip into tools/_prof/ (run once here with `build`, then on the GPU box without arguments)."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANT = os.environ.get("QA_MLP_VARIANT", "")          # "", "W" (no weight loads in the k loop), "A" (no LDS reads), "WA"
SO = os.path.join(ROOT, "tools", "_prof", f"libqa_policy_prof{VARIANT}.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    for v in ("", "W", "A", "WA"):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DQA_MLP_PROF"] +
                              [f"-DQA_MLP_ABLATE_{c}" for c in v] +
                              [os.path.join(ROOT, "quadrupedal_agility_amd", "csrc", "qa_policy.hip"), "-o", SO.replace(f"prof{VARIANT}.so", f"prof{v}.so")])
    sys.exit(0)
import torch
from quadrupedal_agility_amd.rsl_rl.algorithms.fused import PolicyChain
from tests.test_policy_chain import modules
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
lib = C.CDLL(SO)
lib.qa_mlp_pack.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
lib.qa_mlp_forward.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
ac, est, n_obs = modules(seed=1)
ac, est = ac.cuda(), est.cuda()
chain = PolicyChain.describe(ac, est, True)
obs = torch.randn(N, n_obs, device="cuda")
packed = torch.zeros(chain.packed_floats, device="cuda")
w, b = chain._ptr_arrays()
assert lib.qa_mlp_pack(chain.ops, chain.n_ops, w, b, packed.data_ptr(), chain.packed_floats, None) == 0
mean, value, dummy = torch.zeros(N, 12, device="cuda"), torch.zeros(N, 1, device="cuda"), torch.zeros(N, 1, device="cuda")
stamps = torch.zeros(64, dtype=torch.int64, device="cuda")
outs = (C.c_void_p * 4)(mean.data_ptr(), value.data_ptr(), dummy.data_ptr(), stamps.data_ptr())
strides = (C.c_int64 * 4)(12, 1, 1, 1)
for _ in range(5):
    assert lib.qa_mlp_forward(obs.data_ptr(), n_obs, N, n_obs, chain.ops, chain.n_ops, packed.data_ptr(), outs, strides, 4, None) == 0
torch.cuda.synchronize()
s = stamps.cpu().numpy()
t = [s[25]] + list(s[:chain.n_ops + 1])
names = ["stage input"] + [("copy %d" % o.n) if o.kind == 0 else ("layer %d->%d" % (o.k, o.n)) for o in chain.ops]
tot = t[-1] - t[0]
print(f"N={N}: workgroup 0 total {tot} ticks (s_memtime, 100 MHz -> {tot / 100:.1f} us)")
for i, nm in enumerate(names):
    d = t[i + 1] - t[i]
    print(f"  {nm:18s} {d:6d} ticks  {d / 100:6.2f} us")
