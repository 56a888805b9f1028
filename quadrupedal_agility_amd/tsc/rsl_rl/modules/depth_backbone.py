"""Depth encoder of the vision student (tsc/rsl_rl/modules/depth_backbone.py:7-109): `DepthOnlyFCBackbone58x87` (conv 5x5 x 32,
max-pool 2, conv 3x3 x 64, two linears -> 32-d latent) inside `RecurrentDepthBackbone` (latent + proprioception -> MLP -> GRU(512)
-> [latent 32 | delta yaws 2 | softmax obstacle class 6]) with its BYOL head.  Module and parameter names are the reference's.
On a GPU the image stem (both convolutions, the pool and their ELUs, forward and backward) runs on the hand-written kernels of
csrc/qa_conv.hip / qa_gemm.hip through `depth_stem` (r3; MIOpen's per-process kernel search made the iteration time a lottery); the
linears and the GRU stay library calls.  The env-side hot op of this path is the depth ray-cast, csrc/qa_depth.hip."""
import torch
import torch.nn as nn

from . import depth_stem
from .byol import BYOL


class DepthOnlyFCBackbone58x87(nn.Module):
    def __init__(self, prop_dim, scandots_output_dim, hidden_state_dim, output_activation=None, num_frames=1):
        super().__init__()
        self.num_frames = num_frames
        activation = nn.ELU()
        self.image_compression = nn.Sequential(
            nn.Conv2d(in_channels=num_frames, out_channels=32, kernel_size=5),     # [1, 58, 87] -> [32, 54, 83]
            nn.MaxPool2d(kernel_size=2, stride=2),                                 # -> [32, 27, 41]
            activation,
            nn.Conv2d(in_channels=32, out_channels=64, kernel_size=3),             # -> [64, 25, 39]
            activation,
            nn.Flatten(),
            nn.Linear(64 * 25 * 39, 128),
            activation,
            nn.Linear(128, scandots_output_dim))
        self.output_activation = nn.Tanh() if output_activation == "tanh" else activation
        self.augment = None

    def forward(self, images):
        if self.augment:
            images = self.augment(images.clone())
        if depth_stem.ENABLED and images.is_cuda and images.dim() == 3 and depth_stem.stem_matches(self.image_compression):
            return self.output_activation(depth_stem.image_compression(self.image_compression, images))
        return self.output_activation(self.image_compression(images.unsqueeze(1)))


class RecurrentDepthBackbone(nn.Module):
    def __init__(self, base_backbone, n_depth_latent, env_cfg):
        super().__init__()
        self.n_delta_yaw, self.n_obst_type = env_cfg.env.n_delta_yaw, env_cfg.env.n_obst_type
        self.n_depth_latent = n_depth_latent
        activation = nn.ELU()
        self.tanh, self.softmax = nn.Tanh(), nn.Softmax(dim=-1)
        self.base_backbone = base_backbone
        self.byol_learner = BYOL(self.base_backbone, image_size=(58, 87), hidden_layer=-1)
        self.combination_mlp = nn.Sequential(nn.Linear(n_depth_latent + env_cfg.env.n_proprio, 128), activation, nn.Linear(128, n_depth_latent))
        self.rnn = nn.GRU(input_size=n_depth_latent, hidden_size=512, batch_first=True)
        self.output_mlp = nn.Sequential(nn.Linear(512, n_depth_latent + self.n_delta_yaw + self.n_obst_type))
        self.hidden_states = None

    def forward(self, depth_image, proprioception):
        latent = self.combination_mlp(torch.cat((self.base_backbone(depth_image), proprioception), dim=-1))
        latent, self.hidden_states = self.rnn(latent[:, None, :], self.hidden_states)
        out = self.output_mlp(latent.squeeze(1))
        k = self.n_depth_latent + self.n_delta_yaw
        return torch.cat([out[:, :k], self.softmax(out[:, k:])], dim=-1)

    def detach_hidden_states(self):
        self.hidden_states = self.hidden_states.detach().clone()
