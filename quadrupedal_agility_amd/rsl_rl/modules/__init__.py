from .actor_critic import ActorCritic, StateHistoryEncoder, get_activation  # noqa: F401
from .estimator import Estimator  # noqa: F401
