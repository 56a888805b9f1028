#!/usr/bin/env python3
"""tests/golden/tsc_student.npz: the reference's vision-student learner pieces (tsc/rsl_rl: DepthOnlyFCBackbone58x87, RecurrentDepthBackbone,
BYOL, PPO.update_depth_actor, modules/depth_backbone.py, modules/byol.py, algorithms/ppo.py:327-358) run here on CPU through
tests/tsc_student_protocol.py.  Build container only (needs /root/reference).  torchvision is not installed: the only thing the reference
takes from it is the GaussianBlur transform of BYOL's default augmentation, which the protocol replaces by the identity on both sides, so a
stub class stands in for the import."""
import os
import sys
import types
from types import SimpleNamespace

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools", "ref_shims"))
sys.path.insert(0, "/root/reference/tsc")
tv, tvt = types.ModuleType("torchvision"), types.ModuleType("torchvision.transforms")


class _GaussianBlur:            # constructed by BYOL.__init__ inside DEFAULT_AUG; it can only run on the constructor's mock batch (whose
    def __init__(self, *a, **k):    # result is thrown away -- that forward exists to create the projector): the protocol swaps the
        pass                        # augmentations for the identity before any pinned quantity is computed

    def __call__(self, x):
        return x


tvt.GaussianBlur = _GaussianBlur
tv.transforms = tvt
sys.modules["torchvision"], sys.modules["torchvision.transforms"] = tv, tvt

import rsl_rl.modules as ref_modules             # noqa: E402
import rsl_rl.algorithms.ppo as ref_ppo          # noqa: E402
from tests import tsc_student_protocol           # noqa: E402

ns = SimpleNamespace(DepthOnlyFCBackbone58x87=ref_modules.DepthOnlyFCBackbone58x87, RecurrentDepthBackbone=ref_modules.RecurrentDepthBackbone,
                     ActorCriticTSC=ref_modules.ActorCriticTSC, ActorCriticBBC=ref_modules.ActorCriticBBC, Estimator=ref_modules.Estimator, PPO=ref_ppo.PPO)
out = tsc_student_protocol.run(ns)
path = os.path.join(ROOT, "tests", "golden", "tsc_student.npz")
np.savez_compressed(path, **out)
print(path, os.path.getsize(path), "bytes;", {k: getattr(v, "shape", ()) for k, v in out.items()})
