from .rollout_storage import RolloutStorage  # noqa: F401
from .replay_buffer import ReplayBuffer  # noqa: F401
