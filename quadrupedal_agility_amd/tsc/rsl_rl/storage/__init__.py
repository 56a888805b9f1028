from .rollout_storage import RolloutStorage  # noqa: F401
