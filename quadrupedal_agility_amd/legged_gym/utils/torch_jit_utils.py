"""xyzw quaternion / heading helpers with the names the reference code base uses
(bbc/legged_gym/utils/torch_jit_utils.py and the isaacgym.torch_utils functions it star-imports).
On the hot path these are evaluated inside the HIP kernel; the torch versions here serve the mocap
pre-processing (MotionLoader) and host-side tooling.  Standard formulas; batch dimension first."""
import math

import torch


def normalize(x, eps: float = 1e-9):
    return x / x.norm(p=2, dim=-1).clamp(min=eps, max=None).unsqueeze(-1)


def quat_rotate(q, v):
    w = q[:, 3:4]
    qv = q[:, :3]
    return v * (2.0 * w * w - 1.0) + torch.cross(qv, v, dim=-1) * w * 2.0 + qv * (qv * v).sum(-1, keepdim=True) * 2.0


def quat_rotate_inverse(q, v):
    w = q[:, 3:4]
    qv = q[:, :3]
    return v * (2.0 * w * w - 1.0) - torch.cross(qv, v, dim=-1) * w * 2.0 + qv * (qv * v).sum(-1, keepdim=True) * 2.0


def quat_mul(a, b):
    x1, y1, z1, w1 = a.unbind(-1)
    x2, y2, z2, w2 = b.unbind(-1)
    return torch.stack([w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
                        w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2, w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2], dim=-1)


def quat_apply(q, v):
    shape = v.shape
    q = q.reshape(-1, 4)
    v = v.reshape(-1, 3)
    t = torch.cross(q[:, :3], v, dim=-1) * 2
    return (v + q[:, 3:] * t + torch.cross(q[:, :3], t, dim=-1)).view(shape)


def quat_from_angle_axis(angle, axis):
    half = (angle / 2).unsqueeze(-1)
    return normalize(torch.cat([normalize(axis) * half.sin(), half.cos()], dim=-1))


def quat_from_euler_xyz(roll, pitch, yaw):
    cy, sy = torch.cos(yaw * 0.5), torch.sin(yaw * 0.5)
    cr, sr = torch.cos(roll * 0.5), torch.sin(roll * 0.5)
    cp, sp = torch.cos(pitch * 0.5), torch.sin(pitch * 0.5)
    return torch.stack([cy * sr * cp - sy * cr * sp, cy * cr * sp + sy * sr * cp, sy * cr * cp - cy * sr * sp,
                        cy * cr * cp + sy * sr * sp], dim=-1)


def normalize_angle(x):
    return torch.atan2(torch.sin(x), torch.cos(x))


def wrap_to_pi(angles):
    angles = angles % (2 * math.pi)
    return angles - 2 * math.pi * (angles > math.pi)


def torch_rand_float(lower, upper, shape, device):
    return (upper - lower) * torch.rand(*shape, device=device) + lower


def to_torch(x, dtype=torch.float, device="cuda:0", requires_grad=False):
    return torch.tensor(x, dtype=dtype, device=device, requires_grad=requires_grad)


def get_axis_params(value, axis_idx, x_value=0.0, n_dims=3):
    out = [0.0] * n_dims
    out[axis_idx] = value
    out[0] = x_value if axis_idx != 0 else value
    return out


def calc_heading(q):
    ref = torch.zeros_like(q[..., 0:3])
    ref[..., 0] = 1
    d = quat_rotate(q, ref)
    return torch.atan2(d[..., 1], d[..., 0])


def calc_heading_quat_inv(q):
    axis = torch.zeros_like(q[..., 0:3])
    axis[..., 2] = 1
    return quat_from_angle_axis(-calc_heading(q), axis)


def quat_apply_yaw(quat, vec):
    qy = quat.clone().view(-1, 4)
    qy[:, :2] = 0.0
    return quat_apply(normalize(qy), vec)


def euler_from_quaternion(quat_angle):
    x, y, z, w = quat_angle[:, 0], quat_angle[:, 1], quat_angle[:, 2], quat_angle[:, 3]
    roll = torch.atan2(2.0 * (w * x + y * z), 1.0 - 2.0 * (x * x + y * y))
    pitch = torch.asin(torch.clip(2.0 * (w * y - z * x), -1, 1))
    yaw = torch.atan2(2.0 * (w * z + x * y), 1.0 - 2.0 * (y * y + z * z))
    return roll, pitch, yaw


def compute_flat_key_pos(root_states, key_body_pos):
    """Key-body (foot) positions relative to the root, rotated into the heading frame: (N,13),(N,K,3) -> (N,3K)
    (bbc/legged_gym/envs/base/legged_robot.py:1377-1396)."""
    n, k = key_body_pos.shape[0], key_body_pos.shape[1]
    hq = calc_heading_quat_inv(root_states[:, 3:7]).unsqueeze(1).expand(n, k, 4).reshape(n * k, 4)
    rel = (key_body_pos - root_states[:, 0:3].unsqueeze(1)).reshape(n * k, 3)
    return quat_rotate(hq, rel).view(n, k * 3)
