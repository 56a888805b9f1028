from .vec_env import VecEnv  # noqa: F401
