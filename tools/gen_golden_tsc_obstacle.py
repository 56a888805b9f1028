#!/usr/bin/env python3
"""tests/golden/tsc_obstacle.npz: the reference's own `Obstacle` (tsc/legged_gym/utils/obstacle.py) run here for a few seeds
and env counts -- height map, edge mask, goals, obstacle types / origins / yaws / joint positions.  Build container only.
`random` and `numpy.random` are seeded (the class draws from the module-level generators); isaacgym's SubTerrain is the shim
of tools/ref_shims (a zero int16 array with scales); scikit-image is not installed, so `skimage.draw.polygon` is this build's
`fill_polygon` (inside-or-on-boundary pixels within the clipped bounding box): the fixture pins everything the reference
computes around it -- shapes, truncations, rotations, goal transforms, the below-ground marking of the movable parts -- not the
rasterisation rule itself."""
import os
import random
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools", "ref_shims"))
from quadrupedal_agility_amd.tsc.legged_gym.utils.obstacle import fill_polygon      # noqa: E402
from quadrupedal_agility_amd.legged_gym.utils.terrain import SubTerrain            # noqa: E402

tb = types.ModuleType("torch.utils.tensorboard"); tb.SummaryWriter = object; sys.modules["torch.utils.tensorboard"] = tb
for name in ("torchvision", "torchvision.transforms", "cv2", "skimage", "skimage.draw"):
    if name not in sys.modules:
        sys.modules[name] = types.ModuleType(name)
sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
sys.modules["skimage"].draw = sys.modules["skimage.draw"]
sys.modules["skimage.draw"].polygon = lambda r, c, shape=None: fill_polygon(r, c, shape)
import isaacgym                                                                     # noqa: E402
tu = types.ModuleType("isaacgym.terrain_utils"); tu.SubTerrain = SubTerrain
sys.modules["isaacgym.terrain_utils"] = tu; isaacgym.terrain_utils = tu
sys.path.insert(0, "/root/reference/tsc")
cwd = os.getcwd(); os.chdir("/root/reference/tsc/legged_gym/scripts")
import legged_gym.envs.base.legged_robot as _ref_lr                                  # noqa: E402,F401  (import order)
from legged_gym.utils.obstacle import Obstacle                                      # noqa: E402
from legged_gym.envs.go2.go2_agility_config import Go2AgilityCfg                    # noqa: E402
os.chdir(cwd)

out = {}
cases = [(4, 11, False), (9, 5, True)]
for k, (n, seed, curr) in enumerate(cases):
    cfg = Go2AgilityCfg.obstacle()
    cfg.curriculum = curr
    random.seed(seed); np.random.seed(seed)
    ob = Obstacle(cfg, n)
    out[f"c{k}_n"] = np.array(n); out[f"c{k}_seed"] = np.array(seed); out[f"c{k}_curriculum"] = np.array(curr)
    for name in ("height_field_raw", "x_edge_mask", "env_goals", "obstacle_types", "obstacle_origins", "obstacle_yaws", "obstacle_joint_pos",
                 "env_origins", "bar_jump_mask", "tire_jump_mask"):
        out[f"c{k}_{name}"] = np.asarray(getattr(ob, name))
out["num_cases"] = np.array(len(cases))
p = os.path.join(ROOT, "tests", "golden", "tsc_obstacle.npz")
np.savez_compressed(p, **out)
print("wrote", p, os.path.getsize(p))
