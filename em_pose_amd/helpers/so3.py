"""
Exponential / logarithm map of SO(3) under the names and with the clamping semantics of `empose/helpers/so3.py`
(`so3_exponential_map`, so3.py:87-131: the SQUARED angle is clamped at `eps`, so vectors shorter than sqrt(eps) = 0.01 rad
do not give the exact small-angle rotation; `so3_log_map`, so3.py:134-170: sin(angle) is clamped away from zero).
Host-side helpers (any torch device); the HIP kernels carry their own copies of these formulas where they need them
(`csrc/metrics.hip` for the clamped map, `csrc/smpl.hip` for the body model's Rodrigues formula).
"""
import torch


def hat(v):
    """(N,3) -> (N,3,3) skew-symmetric matrices with hat(v) w = v x w."""
    if v.dim() != 2 or v.shape[1] != 3:
        raise ValueError('Input tensor shape has to be Nx3.')
    x, y, z = v.unbind(1)
    o = torch.zeros_like(x)
    return torch.stack([o, -z, y, z, o, -x, -y, x, o], dim=1).reshape(-1, 3, 3)


def hat_inv(h):
    """(N,3,3) skew-symmetric -> (N,3)."""
    if h.dim() != 3 or h.shape[1:] != (3, 3):
        raise ValueError('Input has to be a batch of 3x3 Tensors.')
    return torch.stack([h[:, 2, 1], h[:, 0, 2], h[:, 1, 0]], dim=1)


def so3_rotation_angle(R, eps=1e-4, cos_angle=False):
    if R.dim() != 3 or R.shape[1:] != (3, 3):
        raise ValueError('Input has to be a batch of 3x3 Tensors.')
    tr = R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2]
    if bool(((tr < -1.0 - eps) | (tr > 3.0 + eps)).any()):
        raise ValueError('A matrix has trace outside valid range [-1-eps,3+eps].')
    c = ((tr - 1.0) * 0.5).clamp(-1.0, 1.0)
    return c if cos_angle else torch.acos(c)


def so3_exponential_map(log_rot, eps=1e-4):
    skew = hat(log_rot)
    angle = (log_rot * log_rot).sum(1).clamp(min=eps).sqrt()
    a = (torch.sin(angle) / angle)[:, None, None]
    b = ((1.0 - torch.cos(angle)) / (angle * angle))[:, None, None]
    eye = torch.eye(3, dtype=log_rot.dtype, device=log_rot.device)[None]
    return eye + a * skew + b * torch.bmm(skew, skew)


def so3_log_map(R, eps=1e-4):
    phi = so3_rotation_angle(R)
    s = torch.sin(phi)
    denom = s.abs().clamp(min=eps) * torch.sign(s) + (s == 0).to(phi.dtype) * eps
    return hat_inv((phi / (2.0 * denom))[:, None, None] * (R - R.transpose(1, 2)))


def so3_relative_angle(R1, R2, cos_angle=False):
    return so3_rotation_angle(torch.bmm(R1, R2.transpose(1, 2)), cos_angle=cos_angle)
