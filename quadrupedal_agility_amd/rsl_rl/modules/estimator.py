"""Explicit-privileged-state estimator: prop (57) -> [root height, base lin vel] (4).
Parameter names `estimator.{0,2,4}` as in bbc/rsl_rl/modules/estimator.py:12-36."""
import torch
import torch.nn as nn

from .actor_critic import _run, get_activation


class Estimator(nn.Module):
    def __init__(self, input_dim, output_dim, hidden_dims=[256, 128, 64], activation="elu", **kwargs):
        super().__init__()
        self.input_dim, self.output_dim = input_dim, output_dim
        act = get_activation(activation)
        sizes = [input_dim] + list(hidden_dims) + [output_dim]
        layers = []
        for i in range(len(sizes) - 1):
            layers.append(nn.Linear(sizes[i], sizes[i + 1]))
            if i < len(sizes) - 2:
                layers.append(act)
        self.estimator = nn.Sequential(*layers)

    def forward(self, input):
        return _run(self.estimator, input)

    def inference(self, input):
        with torch.no_grad():
            return self.estimator(input)
