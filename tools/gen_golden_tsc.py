#!/usr/bin/env python3
"""tests/golden/tsc_learner.npz: the reference's task-level learner (tsc/rsl_rl: ActorCriticTSC, ActorCriticBBC,
Estimator, hybrid PPO + RolloutStorage) run here on CPU through tests/tsc_protocol.py.  Build container only (needs
/root/reference); a separate process from tools/gen_golden.py because the BBC and TSC trees both call their package
`rsl_rl`.  The depth-camera modules the package imports are never constructed; their imports are satisfied as they are
(torch only)."""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools", "ref_shims"))
sys.path.insert(0, "/root/reference/tsc")
for name in ("torchvision", "torchvision.transforms"):          # imported by the vision modules, not installed, never used here
    if name not in sys.modules:
        try:
            __import__(name)
        except ImportError:
            sys.modules[name] = types.ModuleType(name)
if not hasattr(sys.modules["torchvision"], "transforms"):
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]

import rsl_rl.modules as ref_modules            # noqa: E402
import rsl_rl.algorithms.ppo as ref_ppo          # noqa: E402
from tests import tsc_protocol                   # noqa: E402

out = tsc_protocol.run(ref_modules, ref_ppo)
path = os.path.join(ROOT, "tests", "golden", "tsc_learner.npz")
np.savez_compressed(path, **out)
print(path, os.path.getsize(path), "bytes;", {k: getattr(v, "shape", ()) for k, v in out.items()})
