"""
Stand-in for numpy-quaternion (reference requirements.txt:21), restating the PUBLISHED definitions of the few functions
the reference's evaluation metrics and sample normalisation call (reference eval/metrics.py:153-160,
data/transforms.py:107,116) [upstream-knowledge]; PARITY UNPINNED against the real package, which this image lacks.
Quaternions are plain float64 arrays (..., 4) = (w, x, y, z) here instead of the package's custom dtype.

  from_rotation_vector(r)            q = exp(r / 2) = (cos(|r|/2), r/|r| sin(|r|/2))
  from_rotation_matrix(R)            unit quaternion of a rotation matrix (sign: w >= 0)
  as_rotation_matrix(q)              the usual quadratic form
  rotation_intrinsic_distance(a, b)  geodesic angle between the two rotations, 2 |log(a b^-1)| with the sign of the
                                     double cover chosen for the shorter arc, i.e. 2 atan2(|v|, |w|) of a * conj(b)
"""
import numpy as np


def from_rotation_vector(rot):
    rot = np.asarray(rot, dtype=np.float64)
    half = 0.5 * np.sqrt((rot * rot).sum(-1, keepdims=True))
    with np.errstate(invalid='ignore', divide='ignore'):
        k = np.where(half > 0, np.sin(half) / (2.0 * half), 0.5)   # sin(|r|/2) / |r|
    return np.concatenate([np.cos(half), rot * k], axis=-1)


def as_rotation_matrix(q):
    q = np.asarray(q, dtype=np.float64)
    q = q / np.sqrt((q * q).sum(-1, keepdims=True))
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    R = np.empty(q.shape[:-1] + (3, 3))
    R[..., 0, 0] = 1 - 2 * (y * y + z * z); R[..., 0, 1] = 2 * (x * y - z * w); R[..., 0, 2] = 2 * (x * z + y * w)
    R[..., 1, 0] = 2 * (x * y + z * w); R[..., 1, 1] = 1 - 2 * (x * x + z * z); R[..., 1, 2] = 2 * (y * z - x * w)
    R[..., 2, 0] = 2 * (x * z - y * w); R[..., 2, 1] = 2 * (y * z + x * w); R[..., 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def from_rotation_matrix(R):
    """Largest-eigenvector form (Bar-Itzhack): robust for every rotation."""
    R = np.asarray(R, dtype=np.float64)
    K = np.empty(R.shape[:-2] + (4, 4))
    K[..., 0, 0] = (R[..., 0, 0] - R[..., 1, 1] - R[..., 2, 2]) / 3
    K[..., 1, 1] = (R[..., 1, 1] - R[..., 0, 0] - R[..., 2, 2]) / 3
    K[..., 2, 2] = (R[..., 2, 2] - R[..., 0, 0] - R[..., 1, 1]) / 3
    K[..., 3, 3] = (R[..., 0, 0] + R[..., 1, 1] + R[..., 2, 2]) / 3
    K[..., 0, 1] = K[..., 1, 0] = (R[..., 1, 0] + R[..., 0, 1]) / 3
    K[..., 0, 2] = K[..., 2, 0] = (R[..., 2, 0] + R[..., 0, 2]) / 3
    K[..., 1, 2] = K[..., 2, 1] = (R[..., 2, 1] + R[..., 1, 2]) / 3
    K[..., 0, 3] = K[..., 3, 0] = (R[..., 2, 1] - R[..., 1, 2]) / 3
    K[..., 1, 3] = K[..., 3, 1] = (R[..., 0, 2] - R[..., 2, 0]) / 3
    K[..., 2, 3] = K[..., 3, 2] = (R[..., 1, 0] - R[..., 0, 1]) / 3
    _, vecs = np.linalg.eigh(K)
    v = vecs[..., -1]                       # (x, y, z, w)
    q = np.concatenate([v[..., 3:4], v[..., :3]], axis=-1)
    return q * np.where(q[..., :1] < 0, -1.0, 1.0)


def rotation_intrinsic_distance(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    aw, av = a[..., 0], a[..., 1:]
    bw, bv = b[..., 0], -b[..., 1:]         # conjugate of b
    w = aw * bw - (av * bv).sum(-1)
    v = aw[..., None] * bv + bw[..., None] * av + np.cross(av, bv)
    return 2.0 * np.arctan2(np.sqrt((v * v).sum(-1)), np.abs(w))
