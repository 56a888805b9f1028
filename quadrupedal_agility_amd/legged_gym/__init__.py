"""Host-side mirror of the reference's `legged_gym` package for the Go2 BBC hot path.

Same public names as bbc/legged_gym (task_registry, LeggedRobot, Go2LocomotionCfg ...), but the
environment is backed by the HIP library behind include/qa_sim.h instead of Isaac Gym.
"""
import os

LEGGED_GYM_ROOT_DIR = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
LEGGED_GYM_ENVS_DIR = os.path.join(LEGGED_GYM_ROOT_DIR, "legged_gym", "envs")
