"""BYOL head of the depth encoder (tsc/rsl_rl/modules/byol.py:1-317, itself adapted from lucidrains/byol-pytorch): two augmented
views of a depth batch, online encoder + projector + predictor against an EMA target encoder, loss 2 - 2 cos.  Parameter names
follow the reference (`net`, `online_encoder.projector`, `online_predictor`, `target_encoder`), so `depth_encoder_state_dict`s load
either way.  torchvision is not a dependency: the one transform used, GaussianBlur((3, 3), (0.5, 1.5)), is a 3 x 3 separable
convolution with reflect padding written out below.

Differences that do not change the math: augmentation draws come from torch's generator on the images' device (the reference mixes
Python's `random` and CPU torch draws); the patch painted by `add_background_noise` is drawn on the device."""
import copy
import random

import torch
import torch.nn.functional as F
from torch import nn


def loss_fn(x, y):
    x, y = F.normalize(x, dim=-1, p=2), F.normalize(y, dim=-1, p=2)
    return 2 - 2 * (x * y).sum(dim=-1)


class RandomApply(nn.Module):
    def __init__(self, fn, p):
        super().__init__()
        self.fn, self.p = fn, p

    def forward(self, x):
        return x if random.random() > self.p else self.fn(x)


class GaussianBlur3(nn.Module):
    """torchvision.transforms.GaussianBlur((3, 3), (sigma_lo, sigma_hi)) on (..., H, W): sigma ~ U per call, taps exp(-x^2 / 2 sigma^2)
    at x = -1, 0, 1 normalised, reflect padding, rows then columns"""
    def __init__(self, sigma=(0.5, 1.5)):
        super().__init__()
        self.sigma = sigma

    def forward(self, x):
        sigma = torch.empty(1).uniform_(*self.sigma).item()
        k = torch.exp(-0.5 * (torch.tensor([-1.0, 0.0, 1.0]) / sigma) ** 2)
        k0, k1, k2 = (k / k.sum()).tolist()                     # the fp32 taps, as host scalars
        # the two 3-tap passes written out on shifted views: as F.conv2d this was MIOpen's per-image im2col + GEMM -- two launches per IMAGE,
        # 2,000-5,000 launches and 30-60 ms of a 200 ms iteration whenever the 10 % draw applied it (profiles/r3_student_*)
        y = F.pad(x.reshape(-1, 1, x.shape[-2], x.shape[-1]), (1, 1, 1, 1), mode="reflect")
        y = k0 * y[..., :, :-2] + k1 * y[..., :, 1:-1] + k2 * y[..., :, 2:]
        y = k0 * y[..., :-2, :] + k1 * y[..., 1:-1, :] + k2 * y[..., 2:, :]
        return y.reshape(x.shape)


class EMA:
    def __init__(self, beta):
        self.beta = beta

    def update_average(self, old, new):
        return new if old is None else old * self.beta + (1 - self.beta) * new


def MLP(dim, projection_size, hidden_size=4096):
    """projector / predictor.  The reference's MaybeSyncBatchnorm (:41-43): in a data-parallel run over RCCL the runner converts these
    BatchNorm1d layers with nn.SyncBatchNorm.convert_sync_batchnorm once the module is on its GPU (SyncBatchNorm cannot run the
    constructor's mock forward on the host)."""
    return nn.Sequential(nn.Linear(dim, hidden_size), nn.BatchNorm1d(hidden_size), nn.ReLU(inplace=True), nn.Linear(hidden_size, projection_size))


class NetWrapper(nn.Module):
    """the base net + a projector created at the first forward (its input width is the net's output width); `layer` -1 = the net's
    own output is the representation (the only mode the depth encoder uses)"""
    def __init__(self, net, projection_size, projection_hidden_size, layer=-1):
        super().__init__()
        if layer != -1:
            raise NotImplementedError("hidden-layer hooks are not used by the depth encoder (hidden_layer=-1)")
        self.net, self.layer = net, layer
        self.projector = None
        self.projection_size, self.projection_hidden_size = projection_size, projection_hidden_size

    def forward(self, x, return_projection=True):
        representation = self.net(x)
        if not return_projection:
            return representation
        if self.projector is None:
            self.projector = MLP(representation.shape[1], self.projection_size, self.projection_hidden_size).to(representation)
        return self.projector(representation), representation


class BYOL(nn.Module):
    def __init__(self, net, image_size, hidden_layer=-1, projection_size=256 // 4, projection_hidden_size=4096 // 4, augment_fn=None,
                 augment_fn2=None, moving_average_decay=0.99, use_momentum=True):
        super().__init__()
        self.net = net
        default_aug = nn.Sequential(RandomApply(self.add_background_noise, p=0.1),
                                    RandomApply(lambda x: x + torch.randn_like(x) * 0.02, p=0.1),
                                    RandomApply(lambda x: x * (torch.rand_like(x) > 0.05).float(), p=0.05),
                                    RandomApply(GaussianBlur3((0.5, 1.5)), p=0.1))
        self.augment1 = augment_fn if augment_fn is not None else default_aug
        self.augment2 = augment_fn2 if augment_fn2 is not None else self.augment1
        self.online_encoder = NetWrapper(net, projection_size, projection_hidden_size, layer=hidden_layer)
        self.use_momentum = use_momentum
        self.target_encoder = None
        self.target_ema_updater = EMA(moving_average_decay)
        self.online_predictor = MLP(projection_size, projection_size, projection_hidden_size)
        device = next(net.parameters()).device
        self.to(device)
        self.forward(torch.randn(2, image_size[0], image_size[1], device=device))       # instantiates projector and target encoder

    @staticmethod
    def add_background_noise(x):
        """a random patch (< 1/4 of each side) of the WHOLE batch overwritten with noise or one constant in (-0.5, 0.5) (:222-240)"""
        height, width = x.shape[1], x.shape[2]
        h = torch.randint(1, height // 4, (1,)).item(); w = torch.randint(1, width // 4, (1,)).item()
        top = torch.randint(0, height - h, (1,)).item(); left = torch.randint(0, width - w, (1,)).item()
        if torch.rand(1) < 0.5:
            patch = torch.rand((h, w), device=x.device, dtype=x.dtype) - 0.5
        else:
            patch = torch.zeros((h, w), device=x.device, dtype=x.dtype) + (torch.rand(1).item() - 0.5)
        x[:, top:top + h, left:left + w] = patch
        return x

    def _get_target_encoder(self):
        if self.target_encoder is None:
            self.target_encoder = copy.deepcopy(self.online_encoder)
            for p in self.target_encoder.parameters():
                p.requires_grad = False
        return self.target_encoder

    def reset_moving_average(self):
        self.target_encoder = None

    def update_moving_average(self):
        assert self.use_momentum and self.target_encoder is not None
        with torch.no_grad():
            cur = list(self.online_encoder.parameters()); ma = list(self.target_encoder.parameters())
            beta = self.target_ema_updater.beta
            # old * beta + (1 - beta) * new with the reference's three roundings (byol.py:72-75): a fused `add(alpha=...)` differs in the
            # last bit, and the projector's zero-gradient pre-BatchNorm biases let Adam turn such a bit into a +-lr step
            # (tests/tsc_student_protocol.py) -- the pinned two-step probe holds only with the same arithmetic
            scaled = torch._foreach_mul(cur, 1 - beta)
            torch._foreach_mul_(ma, beta)
            torch._foreach_add_(ma, scaled)

    def forward(self, x, return_embedding=False, return_projection=True):
        assert not (self.training and x.shape[0] == 1), "BatchNorm in the projector needs more than one sample"
        if return_embedding:
            return self.online_encoder(x, return_projection=return_projection)
        images = torch.cat((self.augment1(x.clone()), self.augment2(x.clone())), dim=0)
        online_projections, _ = self.online_encoder(images)
        pred_one, pred_two = self.online_predictor(online_projections).chunk(2, dim=0)
        with torch.no_grad():
            target = self._get_target_encoder() if self.use_momentum else self.online_encoder
            proj_one, proj_two = target(images)[0].detach().chunk(2, dim=0)
        return (loss_fn(pred_one, proj_two) + loss_fn(pred_two, proj_one)).mean()
