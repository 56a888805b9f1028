from quadrupedal_agility_amd.legged_gym.utils.task_registry import task_registry

from .base.legged_robot import LeggedRobot
from .go2.go2_locomotion_config import Go2LocomotionCfg, Go2LocomotionCfgAlgo

import quadrupedal_agility_amd.legged_gym.utils as _utils

_utils.task_registry = task_registry        # `from legged_gym.utils import task_registry` must yield the object, as in the reference
task_registry.register("go2_locomotion", LeggedRobot, Go2LocomotionCfg(), Go2LocomotionCfgAlgo())
