"""-m gpu: the env-step kernels WITH helper wavefronts (r5, DESIGN 4.1c: two more wavefronts per workgroup take the substep's side chains, the
non-foot rows, the history shift and the closing stores) against the one-wavefront kernels of the same build, from the same arena.

Both are held to the oracle by tests/test_hip_parity.py (the default launch picks helpers at these sizes); this test pins them to EACH OTHER,
which is tighter: the roles execute the same source expressions on the same inputs, so after one env step every tensor must agree to rounding
(different instantiations of one function may contract multiply-adds differently: 1e-5 relative to the tensor's scale, outside a flip budget of
0.2 % of the elements; integers exact), for every export mode a runner uses.  QA_ENV_HELPERS is read once per process, hence the two child processes."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, %(root)r)
from tests.oracle_lib import go2_cfg
from quadrupedal_agility_amd.sim import QaSim
n, lean, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
h = QaSim(go2_cfg(n, seed=7)); h.reset_all()
g = torch.Generator().manual_seed(3)
warm = [torch.randn(n, 12, generator=g).cuda() * 0.5 for _ in range(12)]
for a in warm[:-1]:
    h.step(a)                       # robots in motion, contacts of every kind; identical in both children only if the kernels agree -- so:
torch.cuda.synchronize()
np.save(out + ".pre.npy", h.arena.cpu().numpy())
if len(sys.argv) > 4:               # the second child starts its measured step from the FIRST child's arena
    h.arena.copy_(torch.from_numpy(np.load(sys.argv[4])).cuda())
if lean:
    h.lib.qa_set_lean_exports(h.h, lean)
h.step(warm[-1]); torch.cuda.synchronize()
np.savez(out, **{k: v.cpu().numpy() for k, v in h.t.items()})
"""


@pytest.mark.parametrize("n,lean", [(1000, 0), (512, 3), (4096, 1)])
def test_helper_wavefront_kernels_equal_the_one_wavefront_kernels(tmp_path, n, lean):
    outs = {}
    for mode in ("0", "1"):
        out = str(tmp_path / f"helpers{mode}.npz")
        args = [sys.executable, "-c", CHILD % {"root": ROOT}, str(n), str(lean), out]
        if mode == "1":
            args.append(str(tmp_path / "helpers0.npz.pre.npy"))
        subprocess.run(args, check=True, env=dict(os.environ, QA_ENV_HELPERS=mode), timeout=600)
        outs[mode] = np.load(out)
    a, b = outs["0"], outs["1"]
    worst = {}
    for k in a.files:
        x, y = a[k], b[k]
        if x.dtype.kind in "iub":
            assert x.size == 0 or float((x != y).mean()) <= 2e-3, k
        else:
            if not x.size:
                continue
            d = np.abs(x.astype(np.float64) - y.astype(np.float64))
            worst[k] = float(d.max())
            scale = max(1.0, float(np.max(np.abs(x))))
            # rounding-level agreement everywhere, except where a contact sat on its threshold (the one discontinuity of a step): at most 0.2 % of a
            # tensor's elements, the flip budget of the plane protocol in tests/test_hip_parity.py
            assert float((d > 1e-5 * scale).mean()) <= 2e-3, (k, worst[k], scale, float((d > 1e-5 * scale).mean()))
    print("helpers vs one wavefront, worst |diff|:", {k: v for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:6]})
