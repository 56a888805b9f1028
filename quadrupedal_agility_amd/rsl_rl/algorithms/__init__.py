from .gail import SSInfoGAIL  # noqa: F401
from .discriminator import Discriminator  # noqa: F401
