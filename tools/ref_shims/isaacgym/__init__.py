"""Import-time stand-in for the closed-source `isaacgym` package (build container only).

Lets the reference's pure-torch code import so that tools/gen_golden.py can run it on CPU and dump
golden vectors.  Nothing here simulates anything: gymapi/gymtorch/gymutil/terrain_utils are empty
namespaces; torch_utils holds the ten standard xyzw quaternion helpers the reference star-imports
(SURVEY.md section 8c).  Never shipped to the GPU box, never imported by the product or the tests."""
import types

from . import torch_utils  # noqa: F401

gymapi = types.ModuleType("isaacgym.gymapi")
gymtorch = types.ModuleType("isaacgym.gymtorch")
gymutil = types.ModuleType("isaacgym.gymutil")
terrain_utils = types.ModuleType("isaacgym.terrain_utils")
