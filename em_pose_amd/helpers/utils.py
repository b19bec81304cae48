"""
The helpers of `empose/helpers/utils.py` that callers of the LGD path import, under the reference's names
(utils.py:36-56, 105-123, 156-199).  Host-side conveniences; the kernels have their own restatements.
"""
import numpy as np
import torch

from em_pose_amd.eval.helpers import get_model_dir  # noqa: F401  (reference utils.py:36-39)
from em_pose_amd.nn.loss import mask_from_seq_lengths  # noqa: F401  (reference utils.py:105-123)


def create_model_dir(experiment_dir, experiment_id, model_summary, other_summary=None):
    """`<experiment_dir>/<id>-<summary>[-<other>]`, must not exist yet (reference utils.py:42-51)."""
    import os
    name = '{}-{}'.format(experiment_id, model_summary)
    if other_summary:
        name = '{}-{}'.format(name, other_summary)
    model_dir = os.path.join(experiment_dir, name)
    if os.path.exists(model_dir):
        raise ValueError('Model directory already exists {}'.format(model_dir))
    os.makedirs(model_dir)
    return model_dir


def count_parameters(model):
    """Trainable parameters of a module (reference utils.py:54-56)."""
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


def local_to_global(poses, parents, output_format='aa', input_format='aa'):
    """
    Relative joint rotations -> global ones along the kinematic chain (reference utils.py:165-199).
    :param poses: tensor (N, J*3) of axis-angle vectors or (N, J*9) of row-major rotation matrices.
    :param parents: parents[j] = parent joint of j, negative for the root.
    :return: tensor (N, J*3) or (N, J*9), same dtype and device as `poses`.
    """
    from em_pose_amd.data.transforms import matrix_to_rotvec
    from em_pose_amd.eval.metrics import rotvec_to_matrix
    if output_format not in ('aa', 'rotmat') or input_format not in ('aa', 'rotmat'):
        raise ValueError("formats are 'aa' or 'rotmat'")
    p = poses.detach().cpu().numpy().astype(np.float64)
    dof = 3 if input_format == 'aa' else 9
    n_joints = p.shape[-1] // dof
    local = rotvec_to_matrix(p.reshape(-1, n_joints, 3)) if input_format == 'aa' else p.reshape(-1, n_joints, 3, 3)
    out = np.zeros_like(local)
    for j in range(n_joints):
        out[:, j] = local[:, j] if parents[j] < 0 else out[:, parents[j]] @ local[:, j]
    res = matrix_to_rotvec(out).reshape(-1, n_joints * 3) if output_format == 'aa' else out.reshape(-1, n_joints * 9)
    return torch.from_numpy(res).to(dtype=poses.dtype, device=poses.device)


def global_oris_from_pose(pose_root, pose_body, smpl_parents, angle_idxs):
    """Global orientations (N, F, len(angle_idxs)*9) of the selected joints (reference utils.py:156-162)."""
    n, f = pose_root.shape[0], pose_root.shape[1]
    poses = torch.cat([pose_root.reshape(n * f, -1), pose_body.reshape(n * f, -1)], dim=-1)
    glob = local_to_global(poses, smpl_parents, output_format='rotmat').reshape(n, f, -1, 3, 3)
    return glob[:, :, list(angle_idxs)].reshape(n, f, -1)
