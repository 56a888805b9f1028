"""Running statistics + quaternion helpers of bbc/rsl_rl/utils/utils.py.

`Normalizer` objects are pickled into model.pt under the module path `rsl_rl.utils.utils`
(on_policy_runner.py:306-321); quadrupedal_agility_amd/__init__.py aliases that path to this
module so checkpoints round-trip with the reference.

Difference from the reference: the moments live in fp64 numpy exactly as there (so a pickled
object is interchangeable), but `update_torch` accepts device batches and folds their moments in
with ONE small device->host copy of (mean, var) instead of copying the whole batch
(gail.py:526-529 moves 3 x 1228 x 98 floats to the host per discriminator step).
"""
from typing import Tuple

import numpy as np
import torch

_EPS = np.finfo(float).eps * 4.0


class RunningMeanStd(object):
    def __init__(self, epsilon: float = 1e-4, shape: Tuple[int, ...] = ()):
        self.mean = np.zeros(shape, np.float64)
        self.var = np.ones(shape, np.float64)
        self.count = epsilon

    def update(self, arr: np.ndarray) -> None:
        self.update_from_moments(np.mean(arr, axis=0), np.var(arr, axis=0), arr.shape[0])

    def update_from_moments(self, batch_mean, batch_var, batch_count) -> None:
        # parallel-variance merge (Chan et al.), utils.py:65-84
        delta = batch_mean - self.mean
        total = self.count + batch_count
        m2 = self.var * self.count + batch_var * batch_count + np.square(delta) * self.count * batch_count / total
        self.mean = self.mean + delta * batch_count / total
        self.var = m2 / total
        self.count = total


class Normalizer(RunningMeanStd):
    def __init__(self, input_dim, epsilon=1e-4, clip_obs=10.0):
        super().__init__(shape=input_dim)
        self.epsilon = epsilon
        self.clip_obs = clip_obs

    def normalize(self, input):
        return np.clip((input - self.mean) / np.sqrt(self.var + self.epsilon), -self.clip_obs, self.clip_obs)

    def normalize_torch(self, input, device):
        mean = torch.tensor(self.mean, device=device, dtype=torch.float32)
        std = torch.sqrt(torch.tensor(self.var + self.epsilon, device=device, dtype=torch.float32))
        return torch.clamp((input - mean) / std, -self.clip_obs, self.clip_obs)

    def update_torch(self, batches):
        """Fold in one or more device batches; moments are reduced on the device in fp64."""
        stats = []
        for b in batches:
            b64 = b.detach().to(torch.float64)
            stats.append(torch.stack([b64.mean(dim=0), b64.var(dim=0, unbiased=False)]))
        host = torch.stack(stats).cpu().numpy()
        for b, s in zip(batches, host):
            self.update_from_moments(s[0], s[1], b.shape[0])


def quaternion_slerp(q0, q1, fraction, spin=0, shortestpath=True):
    """Batched slerp (xyzw), semantics of utils.py:121-159: endpoints / identical / zero-angle rows return q0 (or q1
    at fraction 1); otherwise the shortest-arc interpolation."""
    d = torch.sum(q0 * q1, dim=-1, keepdim=True)
    at_zero = torch.isclose(fraction, torch.zeros_like(fraction))
    at_one = torch.isclose(fraction, torch.ones_like(fraction))
    same = (torch.abs(torch.abs(d) - 1.0) < _EPS)
    if shortestpath:
        flip = d < 0
        q1 = torch.where(flip, -q1, q1)
        d = torch.where(flip, -d, d)
    angle = torch.acos(torch.clip(d, -1, 1)) + spin * torch.pi
    tiny = torch.abs(angle) < _EPS
    safe = torch.where(tiny, torch.ones_like(angle), angle)
    isin = 1.0 / safe
    blend = q0 * (torch.sin((1.0 - fraction) * safe) * isin) + q1 * (torch.sin(fraction * safe) * isin)
    out = torch.where(same | tiny | at_zero, q0, blend)
    out = torch.where(at_one & ~(same | tiny | at_zero), q1, out)
    return out


class TorchNormalizer:
    """Device-resident twin of `Normalizer`: identical update rule (fp64), but the moments stay on the
    GPU so the 80 discriminator steps per iteration never touch the host.  `to_reference()` /
    `from_reference()` convert to/from the picklable numpy object that model.pt carries."""

    def __init__(self, input_dim, device, epsilon=1e-4, clip_obs=10.0):
        self.device = device
        self.epsilon, self.clip_obs = epsilon, clip_obs
        self.mean = torch.zeros(input_dim, dtype=torch.float64, device=device)
        self.var = torch.ones(input_dim, dtype=torch.float64, device=device)
        self.count = torch.tensor(float(epsilon), dtype=torch.float64, device=device)

    def _fused(self, t):
        if not (self.mean.is_cuda and t.is_cuda and self.mean.shape[0] <= 128):
            return None
        from quadrupedal_agility_amd.rsl_rl.algorithms import fused
        return fused if fused.ENABLED else None

    def normalize_torch(self, input, device=None):
        f = None if input.requires_grad else self._fused(input)
        if f is not None:
            return f.normalizer_apply(input, self.mean, self.var, self.epsilon, self.clip_obs)
        mean = self.mean.to(torch.float32)
        std = torch.sqrt((self.var + self.epsilon).to(torch.float32))
        return torch.clamp((input - mean) / std, -self.clip_obs, self.clip_obs)

    def update_torch(self, batches):
        f = self._fused(batches[0]) if len(batches) <= 4 else None
        if f is not None:
            return f.normalizer_update(self.mean, self.var, self.count, batches)      # one launch for all batches
        for b in batches:
            b64 = b.detach().to(torch.float64)
            bm, bv, n = b64.mean(dim=0), b64.var(dim=0, unbiased=False), float(b.shape[0])
            delta = bm - self.mean
            total = self.count + n
            m2 = self.var * self.count + bv * n + torch.square(delta) * self.count * n / total
            self.mean.add_(delta * n / total)          # in place: recorded launches keep reading the same buffers
            self.var.copy_(m2 / total)
            self.count.copy_(total)

    def update(self, arr):
        self.update_torch([torch.as_tensor(arr, device=self.device)])

    # ---- data-parallel runs: every rank must fold the SAME (global) batch moments, otherwise the replicas normalise the
    # discriminator's inputs differently (and only rank 0's moments reach model.pt).  The local first and second moments of
    # the batches travel in the discriminator step's gradient bucket and come back averaged over the ranks.
    def batch_moments(self, batches):
        """(k, 2, dim) fp32: per batch [mean, mean of squares] of the local rows TAKEN ABOUT THE RUNNING MEAN (identical on every rank
        before the step).  The bucket they travel in is fp32: raw E[x] / E[x^2] rebuilt as E[x^2] - E[x]^2 cancel to ~1e-7 mean^2, i.e. to
        1e-5 of the variance and worse for features whose mean is large against their spread; shifted by the running mean the
        same rounding is relative to the spread itself, and the single-process update (qa_normalizer_update, double) is matched to 1e-6."""
        shift = self.mean.to(torch.float64)
        out = []
        for b in batches:
            d = b.detach().to(torch.float64) - shift
            out.append(torch.stack([d.mean(dim=0), d.square().mean(dim=0)]))
        self._moment_shift = shift.clone()
        return torch.stack(out).to(torch.float32)

    COLD_FOLDS = 3

    def cold(self):
        """the running mean is not yet a useful shift (first folds of a fresh normaliser): the caller sends `batch_moments_exact` through
        a small fp64 all-reduce of its own instead of the fp32 bucket"""
        return getattr(self, "_folds", 0) < self.COLD_FOLDS

    def batch_moments_exact(self, batches):
        """(k, 2, dim) fp64: per batch the raw [mean, mean of squares] of the local rows (for an fp64 collective)"""
        self._moment_shift = torch.zeros_like(self.mean)
        out = []
        for b in batches:
            b64 = b.detach().to(torch.float64)
            out.append(torch.stack([b64.mean(dim=0), b64.square().mean(dim=0)]))
        return torch.stack(out)

    def update_from_batch_moments(self, moments, rows_per_batch):
        """fold k batches given their (rank-averaged) shifted [mean, mean of squares] and their GLOBAL row counts, in order --
        RunningMeanStd.update_from_moments (utils.py:70-84) with the batch variance E[d^2] - E[d]^2, d = x - shift"""
        m = moments.to(torch.float64)
        self._folds = getattr(self, "_folds", 0) + 1
        shift = getattr(self, "_moment_shift", None)
        if shift is None:
            shift = torch.zeros_like(self.mean)
        for i, n in enumerate(rows_per_batch):
            bm = shift + m[i, 0]
            bv = torch.clamp(m[i, 1] - m[i, 0].square(), min=0.0)
            delta = bm - self.mean
            total = self.count + float(n)
            m2 = self.var * self.count + bv * float(n) + torch.square(delta) * self.count * float(n) / total
            self.mean.add_(delta * float(n) / total)
            self.var.copy_(m2 / total)
            self.count.copy_(total)

    def to_reference(self):
        ref = Normalizer(self.mean.shape[0], epsilon=self.epsilon, clip_obs=self.clip_obs)
        ref.mean, ref.var, ref.count = self.mean.cpu().numpy().copy(), self.var.cpu().numpy().copy(), float(self.count.item())
        return ref

    @classmethod
    def from_reference(cls, ref, device):
        out = cls(ref.mean.shape[0], device, epsilon=ref.epsilon, clip_obs=ref.clip_obs)
        out.mean = torch.as_tensor(ref.mean, dtype=torch.float64, device=device).clone()
        out.var = torch.as_tensor(ref.var, dtype=torch.float64, device=device).clone()
        out.count = torch.tensor(float(ref.count), dtype=torch.float64, device=device)
        return out
