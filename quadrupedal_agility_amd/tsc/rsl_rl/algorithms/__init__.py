from .ppo import PPO  # noqa: F401
from .discriminator import Discriminator  # noqa: F401
