"""
Synthetic SMPL-H-shaped body model and synthetic sensor windows.

The licensed SMPL-H asset (`smplh_amass/neutral/model.npz`), the released weights and the EM-POSE recordings cannot be
shipped or downloaded, so every test and benchmark in this repository runs on a model with the same *shape* as SMPL-H
(SURVEY.md 8d): V vertices on a closed degree-6 triangle mesh, 52 joints (22 body joints with the reference's
SMPL_PARENTS, reference configuration.py:118, plus 2x15 hand joints chained from the wrists), 10 shape and 459
pose blend-shape directions, a 30-nnz convex joint regressor and <=4-nnz convex skinning weights.

`make_model()` returns a dict with the same keys and array layouts as the real `model.npz`, so the same loader
(`em_pose_amd.bodymodels.smpl.load_model_npz`) consumes both.
"""
import numpy as np

from em_pose_amd.helpers.configuration import CONSTANTS as C

# SMPL-H hand joint parents: five 3-joint finger chains per hand hanging off wrist 20 (left) / 21 (right).
_HAND_L = [20, 22, 23, 20, 25, 26, 20, 28, 29, 20, 31, 32, 20, 34, 35]
_HAND_R = [21, 37, 38, 21, 40, 41, 21, 43, 44, 21, 46, 47, 21, 49, 50]
SMPLH_PARENTS = list(C.SMPL_PARENTS) + _HAND_L + _HAND_R

# Approximate SMPL rest joint locations (metres; x left, y up, z forward) for the 22 body joints.
_BODY_JOINTS = np.array([
    [0.00, -0.24, 0.03],  # root
    [0.07, -0.33, 0.02], [-0.07, -0.33, 0.02], [0.00, -0.12, 0.00],  # hips, spine1
    [0.10, -0.71, 0.02], [-0.10, -0.71, 0.02], [0.00, 0.02, 0.02],  # knees, spine2
    [0.09, -1.11, -0.02], [-0.09, -1.11, -0.02], [0.00, 0.08, 0.04],  # ankles, spine3
    [0.11, -1.17, 0.10], [-0.11, -1.17, 0.10], [0.00, 0.29, 0.00],  # feet, neck
    [0.08, 0.20, 0.01], [-0.08, 0.20, 0.01], [0.00, 0.36, 0.04],  # collars, head
    [0.17, 0.23, 0.00], [-0.17, 0.23, 0.00], [0.43, 0.22, -0.01], [-0.43, 0.22, -0.01],  # shoulders, elbows
    [0.68, 0.23, -0.01], [-0.68, 0.23, -0.01],  # wrists
])


def torus_grid_faces(nu, nv):
    """Closed triangle mesh on an nu x nv periodic grid: 2*nu*nv faces, every vertex has degree 6."""
    faces = []
    for i in range(nu):
        for j in range(nv):
            a = i * nv + j
            b = ((i + 1) % nu) * nv + j
            c = ((i + 1) % nu) * nv + (j + 1) % nv
            d = i * nv + (j + 1) % nv
            faces.append([a, b, c])
            faces.append([a, c, d])
    return np.asarray(faces, dtype=np.int64)


def _target_joints():
    J = np.zeros((52, 3))
    J[:22] = _BODY_JOINTS
    for side, wrist, base in ((+1.0, 20, 22), (-1.0, 21, 37)):
        for finger in range(5):
            for k in range(3):
                J[base + finger * 3 + k] = J[wrist] + np.array(
                    [side * (0.03 + 0.025 * (k + 1)), 0.02 * (finger - 2), 0.0])
    return J


def make_model(nu=65, nv=106, seed=6890, n_shape=16, dtype=np.float32):
    """
    Build the synthetic model. Defaults give V = 65*106 = 6890 vertices and 13780 faces.
    :return: dict with keys v_template (V,3), f (F,3), shapedirs (V,3,n_shape), posedirs (V,3,459),
      J_regressor (52,V), weights (V,52), kintree_table (2,52).
    """
    rng = np.random.default_rng(seed)
    V = nu * nv
    faces = torus_grid_faces(nu, nv)

    # A torus stretched to body proportions (roughly 0.9 m wide, 1.75 m tall, 0.3 m deep).
    u = (np.arange(nu) / nu * 2.0 * np.pi)[:, None]
    v = (np.arange(nv) / nv * 2.0 * np.pi)[None, :]
    ring = 1.0 + 0.38 * np.cos(u)
    x = 0.36 * ring * np.cos(v)
    y = -0.27 + 0.70 * ring * np.sin(v)
    z = 0.13 * np.sin(u) * np.ones_like(v)
    v_template = np.stack([x, y, z], axis=-1).reshape(V, 3)
    v_template = v_template + rng.normal(0.0, 0.002, size=v_template.shape)

    Jt = _target_joints()

    # Joint regressor: convex combination of the 30 vertices nearest to the target joint location.
    J_regressor = np.zeros((52, V))
    for j in range(52):
        d = np.linalg.norm(v_template - Jt[j], axis=1)
        nn = np.argsort(d)[:30]
        w = rng.uniform(0.2, 1.0, size=30)
        J_regressor[j, nn] = w / w.sum()
    J_rest = J_regressor @ v_template

    # Skinning weights: 4 nearest joints, smooth fall-off, convex, exactly <=4 non-zeros per row.
    d = np.linalg.norm(v_template[:, None, :] - J_rest[None, :, :], axis=-1)  # (V, 52)
    nn = np.argsort(d, axis=1)[:, :4]
    dn = np.take_along_axis(d, nn, axis=1)
    w = np.exp(-(dn / 0.08) ** 2) + 1e-3
    w[w < 0.02 * w.max(axis=1, keepdims=True)] = 0.0
    w = w / w.sum(axis=1, keepdims=True)
    weights = np.zeros((V, 52))
    np.put_along_axis(weights, nn, w, axis=1)

    shapedirs = rng.normal(0.0, 0.01, size=(V, 3, n_shape))
    posedirs = rng.normal(0.0, 0.001, size=(V, 3, 459))

    kintree = np.stack([np.asarray([2 ** 32 - 1] + SMPLH_PARENTS[1:], dtype=np.int64),
                        np.arange(52, dtype=np.int64)])
    return {
        'v_template': v_template.astype(dtype),
        'f': faces.astype(np.int64),
        'shapedirs': shapedirs.astype(dtype),
        'posedirs': posedirs.astype(dtype),
        'J_regressor': J_regressor.astype(dtype),
        'weights': weights.astype(dtype),
        'kintree_table': kintree,
    }


def small_vertex_ids(n_vertices, n=12, seed=12):
    """Sensor sites for a small test mesh (the real VERTEX_IDS need V >= 5431)."""
    rng = np.random.default_rng(seed)
    return sorted(rng.choice(n_vertices, size=n, replace=False).tolist())


def _exp_so3(r):
    """Rodrigues for a batch of rotation vectors (..., 3) -> (..., 3, 3); float64 helper for data generation only."""
    r = np.asarray(r, dtype=np.float64)
    a = np.linalg.norm(r, axis=-1, keepdims=True)
    a = np.maximum(a, 1e-12)
    k = r / a
    K = np.zeros(r.shape[:-1] + (3, 3))
    K[..., 0, 1], K[..., 0, 2] = -k[..., 2], k[..., 1]
    K[..., 1, 0], K[..., 1, 2] = k[..., 2], -k[..., 0]
    K[..., 2, 0], K[..., 2, 1] = -k[..., 1], k[..., 0]
    s, c = np.sin(a)[..., None], np.cos(a)[..., None]
    return np.eye(3) + s * K + (1.0 - c) * (K @ K)


def make_windows(n_windows, n_frames, seed, sensors_fn=None):
    """
    Ground-truth pose/shape/offset windows as described in SURVEY.md 8d / BASELINE.md 3.
    :param sensors_fn: callable(poses (T,66), betas (T,10), offset_r (T,12,3,3), offset_t (T,12,3)) ->
      (pos (T,12,3), ori (T,12,3,3)) evaluating the body model; when given, noisy sensor readings are produced.
    :return: dict of float32 arrays: poses (B,F,66), shapes (B,10), offset_t (B,12,3), offset_r (B,12,3,3) and,
      if `sensors_fn` is given, marker_pos (B,F,36), marker_oris (B,F,108).
    """
    rng = np.random.default_rng(seed)
    B, F = n_windows, n_frames
    base = rng.normal(0.0, 0.2, size=(B, 1, 66))
    walk = np.cumsum(rng.normal(0.0, 0.02, size=(B, F, 66)), axis=1)
    poses = base + walk
    poses[:, :, :3] -= poses[:, :1, :3]  # root of frame 0 is the identity (mimics root normalisation)
    shapes = np.clip(rng.normal(0.0, 1.0, size=(B, 10)), -2.0, 2.0)
    offset_t = rng.normal(0.0, 0.02, size=(B, 12, 3))
    offset_r = _exp_so3(rng.normal(0.0, 0.1, size=(B, 12, 3)))
    out = {'poses': poses.astype(np.float32), 'shapes': shapes.astype(np.float32),
           'offset_t': offset_t.astype(np.float32), 'offset_r': offset_r.astype(np.float32)}
    if sensors_fn is not None:
        T = B * F
        betas = np.repeat(shapes[:, None], F, axis=1).reshape(T, 10)
        o_r = np.repeat(offset_r[:, None], F, axis=1).reshape(T, 12, 3, 3)
        o_t = np.repeat(offset_t[:, None], F, axis=1).reshape(T, 12, 3)
        pos, ori = sensors_fn(poses.reshape(T, 66).astype(np.float32), betas.astype(np.float32),
                              o_r.astype(np.float32), o_t.astype(np.float32))
        pos = np.asarray(pos, dtype=np.float64) + rng.normal(0.0, 0.005, size=(T, 12, 3))
        noise_r = _exp_so3(rng.normal(0.0, np.deg2rad(2.0), size=(T, 12, 3)))
        ori = np.asarray(ori, dtype=np.float64) @ noise_r
        out['marker_pos'] = pos.reshape(B, F, 36).astype(np.float32)
        out['marker_oris'] = ori.reshape(B, F, 108).astype(np.float32)
    return out


# Frame counts of the 36 recordings of the EM-POSE test set (reference README.md:107-142), used to shape the synthetic
# stand-in of BASELINE.json configs[3] (sum = 54 030 frames).
README_SEQUENCE_LENGTHS = [3460, 1937, 688, 2213, 490, 1630, 801, 1945, 1875, 2916, 1311, 745, 1796, 246, 1423, 1331,
                           1647, 1569, 2931, 596, 1421, 1846, 296, 1191, 560, 1736, 1677, 2458, 779, 1269, 2002, 504,
                           1600, 1191, 2303, 1647]


def make_sequence(n_frames, seed, sensors_fn, missing_rate=0.002):
    """One synthetic recording in the shape of a `*_clean.npz` file (reference data.py:161-171)."""
    w = make_windows(1, n_frames, seed, sensors_fn)
    rng = np.random.default_rng(seed + 17)
    masks = (rng.uniform(size=(n_frames, 12)) > missing_rate)
    return {'id': np.asarray('synthetic_%05d' % seed), 'sensor_pos': w['marker_pos'][0].reshape(n_frames, 12, 3),
            'sensor_oris': w['marker_oris'][0].reshape(n_frames, 12, 3, 3), 'sensor_masks': masks,
            'smpl_poses': w['poses'][0], 'smpl_shape': w['shapes'][0], 'smpl_trans': np.zeros((n_frames, 3), np.float32),
            'offset_means': w['offset_t'][0], 'offset_covs': np.tile(np.eye(3, dtype=np.float32) * 1e-4, (12, 1, 1)),
            'offset_r': w['offset_r'][0]}
