from .actor_critic import Actor, ActorCriticBBC, ActorCriticTSC  # noqa: F401
from quadrupedal_agility_amd.rsl_rl.modules.estimator import Estimator  # noqa: F401  (same module as the BBC tree: tsc/rsl_rl/modules/estimator.py)
