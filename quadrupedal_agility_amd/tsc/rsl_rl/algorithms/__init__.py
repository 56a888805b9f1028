from .ppo import PPO  # noqa: F401
