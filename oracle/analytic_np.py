"""
ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/torch_ref.py for the rules).

Second, independent restatement of one SMPL evaluation of the LGD loop, in NumPy (float64 by default), in the
*factored* form the HIP kernels use: sub-mesh only, hand joints folded into the wrists, template + blend-shapes +
joint regression as one matrix product, and a hand-derived reverse pass instead of autograd.  Its job is to

  * prove (tests/test_analytic_vs_autograd.py) that the factored evaluation and its analytic gradient equal the dense
    evaluation + torch.autograd of oracle/torch_ref.py (which is pinned to the reference), and
  * expose every intermediate buffer (feat, out, d_out, d_R ...) so that each HIP kernel can be checked on its own.

Math (per frame; SURVEY.md Appendix A, reference models.py:471-483,560-579, virtual_sensors.py:16-38, loss.py:23-41):
  R_j = rodrigues(theta_j)                       feat = [vec(R_1-I)..vec(R_21-I), beta, 1]
  out = feat @ Wc^T = [v_posed (needed verts) | J]
  G_j = G_parent * [R_j | J_j - J_parent]        joints_j = G_j^t      A_j = [G_j^R | G_j^t - G_j^R J_j]
  v_s = sum_k w_sk (A_b^R vp_s + A_b^t)
  n = mean_f (v_f1 - v_f0) x (v_f2 - v_f0); nh = n/|n|; s = (v_h - v_c)/|.|; t = (nh x s)/|.|; s' = (t x nh)/|.|
  R_m = [s' t nh];  R^_m = R_m R_off;  p^_m = v_c + R_m t_off
  E = scale * sum_{m in idx} (|p^_m - p*_m| + |R^_m - R*_m|_F)
Reverse pass of the chain in world-space form: with F_b = sum w dv_s, M_b = sum w dv_s (x) (A_b^R vp_s + A_b^t) and
subtree sums Fs_j, Ms_j:   dR_j = G_p^RT (Ms_j - Fs_j (x) t_j) G_j^R ,   dJ_j = (G_p^R - G_j^R)^T Fs_j.
"""
import numpy as np


def rodrigues_fwd(theta, eps=1e-8, convention='smplx'):
    """theta (...,3) -> R (...,3,3) and the saved quantities for the reverse pass.  `convention`: 'smplx' guards the
    angle as ||theta + eps||, 'so3' as sqrt(clamp(||theta||^2, 1e-4)) (reference helpers/so3.py:116-121); `u` is
    a * d(a)/d(theta) in either case."""
    if convention == 'smplx':
        u = theta + eps
        a = np.sqrt((u * u).sum(-1, keepdims=True))
    elif convention == 'so3':
        n2 = (theta * theta).sum(-1, keepdims=True)
        a = np.sqrt(np.maximum(n2, 1e-4))
        u = np.where(n2 < 1e-4, 0.0, theta)
    else:
        raise ValueError(convention)
    d = theta / a
    s, c = np.sin(a)[..., None], np.cos(a)[..., None]
    K = np.zeros(theta.shape[:-1] + (3, 3), dtype=theta.dtype)
    K[..., 0, 1], K[..., 0, 2] = -d[..., 2], d[..., 1]
    K[..., 1, 0], K[..., 1, 2] = d[..., 2], -d[..., 0]
    K[..., 2, 0], K[..., 2, 1] = -d[..., 1], d[..., 0]
    R = np.eye(3, dtype=theta.dtype) + s * K + (1.0 - c) * (K @ K)
    return R, (u, a, d, K, s, c)


def rodrigues_bwd(dR, saved):
    """dE/dR (...,3,3) -> dE/dtheta (...,3)."""
    u, a, d, K, s, c = saved
    KK = K @ K
    ds = (dR * K).sum((-1, -2))[..., None]
    dc1 = (dR * KK).sum((-1, -2))[..., None]  # w.r.t. (1 - cos a)
    Kt = np.swapaxes(K, -1, -2)
    dK = s * dR + (1.0 - c) * (dR @ Kt + Kt @ dR)
    dd = np.stack([dK[..., 2, 1] - dK[..., 1, 2], dK[..., 0, 2] - dK[..., 2, 0], dK[..., 1, 0] - dK[..., 0, 1]], -1)
    da = ds * c[..., 0] + dc1 * s[..., 0]
    dtheta = dd / a
    da = da - (dd * d).sum(-1, keepdims=True) / a
    return dtheta + da * u / a


def _unit(x):
    n = np.sqrt((x * x).sum(-1, keepdims=True))
    return x / n, n


def _unit_bwd(dy, y, n):
    return (dy - y * (dy * y).sum(-1, keepdims=True)) / n


def features(theta, beta, convention='smplx'):
    T = theta.shape[0]
    R, saved = rodrigues_fwd(theta.reshape(T, 22, 3), convention=convention)
    feat = np.zeros((T, 200), dtype=theta.dtype)
    feat[:, :189] = (R[:, 1:] - np.eye(3, dtype=theta.dtype)).reshape(T, 189)
    feat[:, 189:199] = beta
    feat[:, 199] = 1.0
    return R, saved, feat


def chain_fwd(tab, R, J):
    T = R.shape[0]
    parents = tab['parents']
    GR = np.zeros((T, 22, 3, 3), dtype=R.dtype)
    Gt = np.zeros((T, 22, 3), dtype=R.dtype)
    GR[:, 0], Gt[:, 0] = R[:, 0], J[:, 0]
    for j in range(1, 22):
        p = parents[j]
        GR[:, j] = GR[:, p] @ R[:, j]
        Gt[:, j] = np.einsum('tab,tb->ta', GR[:, p], J[:, j] - J[:, p]) + Gt[:, p]
    At = Gt - np.einsum('tjab,tjb->tja', GR, J)
    return GR, Gt, At


def skin_fwd(tab, GR, At, vp):
    idx, w = tab['skin_idx'], tab['skin_w'].astype(vp.dtype)
    v = np.zeros_like(vp)
    for k in range(idx.shape[1]):
        b = idx[:, k]
        v += w[None, :, k, None] * (np.einsum('tsab,tsb->tsa', GR[:, b], vp) + At[:, b])
    return v


def sensors_fwd(tab, v):
    M = tab['n_sensors']
    T = v.shape[0]
    n = np.zeros((T, M, 3), dtype=v.dtype)
    for m in range(M):
        for k in range(tab['s_deg'][m]):
            f = tab['s_faces'][m, k]
            n[:, m] += np.cross(v[:, f[1]] - v[:, f[0]], v[:, f[2]] - v[:, f[0]])
        n[:, m] /= tab['s_deg'][m]
    vc, vh = v[:, tab['s_center']], v[:, tab['s_helper']]
    nh, n_n = _unit(n)
    s, n_e = _unit(vh - vc)
    t, n_b = _unit(np.cross(nh, s))
    s2, n_a = _unit(np.cross(t, nh))
    Rm = np.stack([s2, t, nh], axis=-1)
    return vc, Rm, n, (nh, n_n, s, n_e, t, n_b, s2, n_a)


def sensors_bwd(tab, v, saved, dvc, dRm):
    """Cotangents of (centre position, frame) -> dv over the needed vertices."""
    nh, n_n, s, n_e, t, n_b, s2, n_a = saved
    M = tab['n_sensors']
    dv = np.zeros_like(v)
    ds2, dt, dnh = dRm[..., 0].copy(), dRm[..., 1].copy(), dRm[..., 2].copy()
    da = _unit_bwd(ds2, s2, n_a)  # a = t x nh
    dt += np.cross(nh, da)
    dnh += np.cross(da, t)
    db = _unit_bwd(dt, t, n_b)  # b = nh x s
    dnh += np.cross(s, db)
    ds = np.cross(db, nh)
    de = _unit_bwd(ds, s, n_e)
    dn = _unit_bwd(dnh, nh, n_n)
    for m in range(M):
        c, h = tab['s_center'][m], tab['s_helper'][m]
        dv[:, c] += dvc[:, m] - de[:, m]
        dv[:, h] += de[:, m]
        dfn = dn[:, m] / tab['s_deg'][m]
        for k in range(tab['s_deg'][m]):
            f = tab['s_faces'][m, k]
            e1, e2 = v[:, f[1]] - v[:, f[0]], v[:, f[2]] - v[:, f[0]]
            de1, de2 = np.cross(e2, dfn), np.cross(dfn, e1)
            dv[:, f[1]] += de1
            dv[:, f[2]] += de2
            dv[:, f[0]] -= de1 + de2
    return dv


def frame_scale(seq_lengths, marker_masks, B, F, dtype=np.float64):
    """Per-frame loss weight after the reference's `* batch_size * seq_length` (models.py:578-579, loss.py:31-41)."""
    sl = np.asarray(seq_lengths).reshape(B).astype(dtype)
    live = (np.arange(F)[None, :] < sl[:, None]).astype(dtype)
    scale = live * (F / sl)[:, None]
    if marker_masks is not None:
        scale = scale * np.all(np.asarray(marker_masks).reshape(B, F, -1) != 0, axis=-1)
    return scale.reshape(B * F)


def smpl_sensors(tab, theta, beta, off_r, off_t, tgt_pos=None, tgt_ori=None, idx=None, scale=None,
                 convention='smplx'):
    """
    One evaluation (+ optionally the residual gradient).
    :param theta (T,66) beta (T,10) off_r (T,12,3,3) off_t (T,12,3)
    :param tgt_pos (T,Mu,3), tgt_ori (T,Mu,3,3): measured sensors for the `idx` subset; scale (T,) frame weights.
    :return: dict with pos (T,12,3), ori (T,12,3,3), joints (T,22,3) and, if targets are given, g_theta, g_beta and
      the intermediates feat, out, d_out, d_feat, d_R_chain.
    """
    dt = theta.dtype
    T = theta.shape[0]
    wc = tab['wc'].astype(dt)
    nv, j_off = tab['nv'], tab['j_off']
    R, rsaved, feat = features(theta, beta, convention)
    out = feat @ wc.T
    vp = out[:, :nv * 3].reshape(T, nv, 3)
    J = out[:, j_off:j_off + 66].reshape(T, 22, 3)
    GR, Gt, At = chain_fwd(tab, R, J)
    v = skin_fwd(tab, GR, At, vp)
    vc, Rm, n, ssaved = sensors_fwd(tab, v)
    ori = Rm @ off_r
    pos = vc + np.einsum('tmab,tmb->tma', Rm, off_t)
    res = {'pos': pos, 'ori': ori, 'joints': Gt, 'feat': feat, 'out': out, 'R': R, 'v': v, 'normals': n}
    if tgt_pos is None:
        return res

    idx = list(range(12)) if idx is None else list(idx)
    scale = np.ones(T, dtype=dt) if scale is None else scale.astype(dt)
    dpos = np.zeros_like(pos)
    dori = np.zeros_like(ori)
    rp = pos[:, idx] - tgt_pos
    ro = ori[:, idx] - tgt_ori
    with np.errstate(invalid='ignore', divide='ignore'):
        gp = rp / np.sqrt((rp * rp).sum(-1, keepdims=True))
        go = ro / np.sqrt((ro * ro).sum((-1, -2), keepdims=True))
    # a masked/padded frame contributes exactly zero (its inputs may be all-zero: 0 * nan guarded)
    live = scale != 0
    dpos[:, idx] = np.where(live[:, None, None], gp * scale[:, None, None], 0.0)
    dori[:, idx] = np.where(live[:, None, None, None], go * scale[:, None, None, None], 0.0)
    res['energy'] = scale * (np.sqrt((rp * rp).sum(-1)).sum(-1) + np.sqrt((ro * ro).sum((-1, -2))).sum(-1))

    dRm = dori @ np.swapaxes(off_r, -1, -2) + dpos[..., :, None] * off_t[..., None, :]
    dv = sensors_bwd(tab, v, ssaved, dpos, dRm)

    # skinning reverse
    sidx, sw = tab['skin_idx'], tab['skin_w'].astype(dt)
    dvp = np.zeros_like(vp)
    Fb = np.zeros((T, 22, 3), dtype=dt)
    Mb = np.zeros((T, 22, 3, 3), dtype=dt)
    for k in range(sidx.shape[1]):
        b = sidx[:, k]
        wk = sw[None, :, k, None]
        dvp += wk * np.einsum('tsba,tsb->tsa', GR[:, b], dv)
        x = np.einsum('tsab,tsb->tsa', GR[:, b], vp) + At[:, b]
        for s in range(nv):
            Fb[:, b[s]] += sw[s, k] * dv[:, s]
            Mb[:, b[s]] += sw[s, k] * dv[:, s, :, None] * x[:, s, None, :]
    # subtree sums and the world-space chain gradient
    parents = tab['parents']
    dR = np.zeros((T, 22, 3, 3), dtype=dt)
    dJ = np.zeros((T, 22, 3), dtype=dt)
    eye = np.broadcast_to(np.eye(3, dtype=dt), (T, 3, 3))
    for j in range(22):
        members = tab['sub'][tab['sub_ptr'][j]:tab['sub_ptr'][j + 1]]
        Fs = Fb[:, members].sum(1)
        Ms = Mb[:, members].sum(1)
        X = Ms - Fs[:, :, None] * Gt[:, j, None, :]
        Gp = GR[:, parents[j]] if parents[j] >= 0 else eye
        dR[:, j] = np.swapaxes(Gp, -1, -2) @ X @ GR[:, j]
        dJ[:, j] = np.einsum('tab,ta->tb', Gp - GR[:, j], Fs)
    d_out = np.zeros_like(out)
    d_out[:, :nv * 3] = dvp.reshape(T, -1)
    d_out[:, j_off:j_off + 66] = dJ.reshape(T, -1)
    d_feat = d_out @ wc
    dR_total = dR.copy()
    dR_total[:, 1:] += d_feat[:, :189].reshape(T, 21, 3, 3)
    g_theta = rodrigues_bwd(dR_total, rsaved).reshape(T, 66)
    g_beta = d_feat[:, 189:199]
    res.update({'g_theta': g_theta, 'g_beta': g_beta, 'd_out': d_out, 'd_feat': d_feat, 'd_R_chain': dR, 'dv': dv})
    return res
