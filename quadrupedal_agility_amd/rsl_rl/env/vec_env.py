"""Duck type the runner drives (bbc/rsl_rl/env/vec_env.py:7-36)."""
from abc import ABC, abstractmethod
from typing import Tuple, Union

import torch


class VecEnv(ABC):
    num_envs: int
    num_obs: int
    num_privileged_obs: int
    num_actions: int
    max_episode_length: int
    privileged_obs_buf: torch.Tensor
    obs_buf: torch.Tensor
    rew_buf: torch.Tensor
    reset_buf: torch.Tensor
    episode_length_buf: torch.Tensor
    extras: dict
    device: torch.device

    @abstractmethod
    def step(self, actions: torch.Tensor) -> Tuple[torch.Tensor, Union[torch.Tensor, None], torch.Tensor, torch.Tensor, dict]:
        ...

    @abstractmethod
    def reset(self, env_ids: Union[list, torch.Tensor]):
        ...

    @abstractmethod
    def get_observations(self) -> torch.Tensor:
        ...

    @abstractmethod
    def get_privileged_observations(self) -> Union[torch.Tensor, None]:
        ...
