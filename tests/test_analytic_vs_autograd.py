"""
The factored sub-mesh evaluation + hand-derived reverse pass (oracle/analytic_np.py, the blueprint of the HIP
kernels, built on the product's packed tables) must equal the dense full-mesh evaluation + torch.autograd of
oracle/torch_ref.py (pinned to the reference).  float64, CPU.
"""
import numpy as np
import pytest
import torch

from em_pose_amd import synthetic
from em_pose_amd.bodymodels import tables as TB
from oracle import analytic_np as A
from oracle import torch_ref as R
from tests import helpers as H


def _setup(model, vids, T, seed):
    rng = np.random.default_rng(seed)
    theta = rng.normal(0, 0.25, size=(T, 66))
    beta = rng.normal(0, 1.0, size=(T, 10))
    off_t = rng.normal(0, 0.02, size=(T, 12, 3))
    off_r = synthetic._exp_so3(rng.normal(0, 0.1, size=(T, 12, 3)))
    return theta, beta, off_r, off_t


@pytest.mark.parametrize('idx,conv', [(None, 'smplx'), (R.S_CONFIG_6, 'smplx'), (None, 'so3'), (R.S_CONFIG_6, 'so3')])
def test_small_model_matches_autograd(idx, conv):
    """Both Rodrigues conventions (angle guard ||r + 1e-8|| vs the clamp of reference helpers/so3.py:116-121), with
    joints at exactly zero, below and above the clamp."""
    model = H.small_model()
    vids = synthetic.small_vertex_ids(160)
    tab = TB.build_lgd_tables(model, vids, dtype=np.float64)
    T = 6
    theta, beta, off_r, off_t = _setup(model, vids, T, 3)
    theta[0, 9:30] = 0.0
    theta[1, 3:] *= 1e-2
    theta[2, 6:9] = 0.0099 / np.sqrt(3)
    theta[2, 9:12] = 0.0101 / np.sqrt(3)
    rng = np.random.default_rng(9)
    ids = list(range(12)) if idx is None else idx

    bm = R.BodyModelTensors(model, dtype=torch.float64, rodrigues_convention=conv)
    tables = R.sensor_tables(model['f'], vids)
    th = torch.from_numpy(theta).requires_grad_(True)
    be = torch.from_numpy(beta).requires_grad_(True)
    pos, ori, joints = R.estimated_markers(bm, tables, vids, th, be, torch.from_numpy(off_r), torch.from_numpy(off_t))
    tgt_pos = pos.detach().numpy()[:, ids] + rng.normal(0, 0.01, size=(T, len(ids), 3))
    tgt_ori = ori.detach().numpy()[:, ids] @ synthetic._exp_so3(rng.normal(0, 0.05, size=(T, len(ids), 3)))
    scale = np.array([1.0, 0.5, 2.0, 0.0, 1.0, 1.5])
    e = (torch.from_numpy(scale) * (
        torch.sqrt(((pos[:, ids] - torch.from_numpy(tgt_pos)) ** 2).sum(-1)).sum(-1) +
        torch.sqrt(((ori[:, ids] - torch.from_numpy(tgt_ori)) ** 2).sum((-1, -2))).sum(-1))).sum()
    g_th, g_be = torch.autograd.grad(e, [th, be])

    res = A.smpl_sensors(tab, theta, beta, off_r, off_t, tgt_pos, tgt_ori, ids, scale, convention=conv)
    np.testing.assert_allclose(res['pos'], pos.detach().numpy(), atol=1e-12)
    np.testing.assert_allclose(res['ori'], ori.detach().numpy(), atol=1e-11)
    np.testing.assert_allclose(res['joints'], joints.detach().numpy(), atol=1e-12)
    np.testing.assert_allclose(res['g_theta'], g_th.numpy(), atol=1e-9)
    np.testing.assert_allclose(res['g_beta'], g_be.numpy(), atol=1e-9)


def test_finite_differences():
    model = H.small_model()
    vids = synthetic.small_vertex_ids(160)
    tab = TB.build_lgd_tables(model, vids)
    theta, beta, off_r, off_t = _setup(model, vids, 2, 4)
    base = A.smpl_sensors(tab, theta, beta, off_r, off_t)
    rng = np.random.default_rng(1)
    tgt_pos = base['pos'] + rng.normal(0, 0.01, size=base['pos'].shape)
    tgt_ori = base['ori'] + rng.normal(0, 0.02, size=base['ori'].shape)
    res = A.smpl_sensors(tab, theta, beta, off_r, off_t, tgt_pos, tgt_ori)
    h = 1e-6
    for (arr, g) in ((theta, res['g_theta']), (beta, res['g_beta'])):
        for col in rng.choice(arr.shape[1], size=8, replace=False):
            p, m = arr.copy(), arr.copy()
            p[:, col] += h
            m[:, col] -= h
            args_p = (p, beta) if arr is theta else (theta, p)
            args_m = (m, beta) if arr is theta else (theta, m)
            ep = A.smpl_sensors(tab, *args_p, off_r, off_t, tgt_pos, tgt_ori)['energy']
            em = A.smpl_sensors(tab, *args_m, off_r, off_t, tgt_pos, tgt_ori)['energy']
            np.testing.assert_allclose((ep - em) / (2 * h), g[:, col], rtol=1e-5, atol=1e-6)


def test_invariants():
    """theta=0,beta=0 -> template; rigid root rotation rotates everything; hand folding is exact."""
    model = H.small_model()
    bm = R.BodyModelTensors(model, dtype=torch.float64)
    z = torch.zeros(1, 63, dtype=torch.float64)
    v, j = R.smpl_fk(bm, z, torch.zeros(1, 10, dtype=torch.float64))
    np.testing.assert_allclose(v[0].numpy(), model['v_template'].astype(np.float64), atol=1e-12)
    np.testing.assert_allclose(j[0].numpy(), model['J_regressor'].astype(np.float64) @ model['v_template'], atol=1e-12)

    rng = np.random.default_rng(0)
    pose = torch.from_numpy(rng.normal(0, 0.3, size=(1, 63)))
    beta = torch.from_numpy(rng.normal(0, 1, size=(1, 10)))
    root = torch.from_numpy(rng.normal(0, 0.5, size=(1, 3)))
    v0, j0 = R.smpl_fk(bm, pose, beta)
    v1, j1 = R.smpl_fk(bm, pose, beta, root)
    Rr = R.rodrigues(root)[0]
    np.testing.assert_allclose((v1[0] - j1[0, 0]).numpy(), ((v0[0] - j0[0, 0]) @ Rr.T).numpy(), atol=1e-7)  # the +1e-8 in the angle makes R orthonormal only to ~1e-8

    # folded 22-joint skinning == dense 52-joint skinning on every vertex (full-mesh tables)
    full = TB.build_full_mesh_tables(model, dtype=np.float64)
    theta = np.concatenate([root.numpy(), pose.numpy()], axis=1)
    Rm, _, feat = A.features(theta, beta.numpy())
    out = feat @ full['wc'].astype(np.float64).T
    V = full['n_vertices']
    vp = out[:, :V * 3].reshape(1, V, 3)
    J = out[:, full['j_off']:full['j_off'] + 66].reshape(1, 22, 3)
    GR, Gt, At = A.chain_fwd(full, Rm, J)
    vfold = A.skin_fwd(full, GR, At, vp)
    np.testing.assert_allclose(vfold[0], v1[0].numpy(), atol=1e-11)
    np.testing.assert_allclose(Gt[0], j1[0, :22].numpy(), atol=1e-11)


def test_big_model_tables_shapes():
    model = synthetic.make_model()
    from em_pose_amd.helpers.configuration import CONSTANTS as C
    tab = TB.build_lgd_tables(model, C.VERTEX_IDS)
    assert tab['n_sensors'] == 12 and tab['kb'] <= 4
    assert 60 <= tab['nv'] <= 120 and tab['max_deg'] == 6
    assert tab['wc'].shape == (tab['ncp'], 200) and tab['ncp'] % 4 == 0
    assert tab['bone_ptr'][-1] == len(tab['bone_vert'])
    np.testing.assert_allclose(tab['skin_w'].sum(1), 1.0, atol=1e-6)
