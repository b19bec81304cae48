"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Nothing in `em_pose_amd/` may import this module; only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg use it, as the checker / the timed CPU baseline.

A plain-PyTorch (CPU, dtype-parametric) restatement of the reference's LGD path, written to follow the reference's
*algorithm* step for step -- dense full-mesh SMPL-H skinning, index-gather virtual sensors, and `autograd` for the
in-loop residual gradient -- so that (a) it can be checked 1:1 against the imported reference in the build container
(tests/golden/make_golden.py) and (b) timing it is a faithful proxy for the reference's CPU path.

Parity status
-------------
* Everything the reference itself implements (a1-a15 of SURVEY.md section 8) is PINNED: the committed fixtures under
  tests/golden/ were produced by the unmodified reference modules imported from /root/reference, and
  tests/test_oracle_golden.py checks this restatement against them.
* The SMPL-H body model arithmetic (`body_model_forward`) is **parity unpinned**: it lives in the un-vendored
  dependency `human-body-prior` (fork github.com/totomobile43/human_body_prior @ 821a0e7, reference
  requirements.txt:9; call sites reference empose/bodymodels/smpl.py:42,121-122) whose source is not available.
  It restates the published SMPL / smplx `lbs` algorithm (Loper et al. 2015; smplx lbs.py) and is anchored on the
  reference's call sites and on analytic invariants (tests/test_analytic_vs_autograd.py::test_invariants).
"""
import numpy as np
import torch

N_BODY = 22  # root + 21 body joints; reference configuration.py:104


# ----------------------------------------------------------------------------------------------------------------------
# Mesh topology helpers (restating what the reference gets from trimesh==3.9.32, reference smpl.py:58-67,
# virtual_sensors.py:47-75).  [upstream-knowledge]: `Trimesh.vertex_faces` = `geometry.vertex_face_indices`, whose rows
# are filled from `faces_sparse.dot(identity).nonzero()[1]`: scipy's CSR product lists the columns of a row in reverse
# insertion order, i.e. DESCENDING face id (trimesh's slow-loop fallback reverses explicitly to match).  The stand-in
# oracle/refstubs/trimesh evaluates that scipy expression itself; this function states the resulting order directly.
# ----------------------------------------------------------------------------------------------------------------------
def vertex_faces_table(faces, n_vertices):
    faces = np.asarray(faces, dtype=np.int64)
    flat = faces.reshape(-1)
    counts = np.bincount(flat, minlength=n_vertices)
    order = np.argsort(flat, kind='stable')  # face-major scan => ascending face ids per vertex
    table = -np.ones((n_vertices, max(int(counts.max()), 1)), dtype=np.int64)
    starts = np.concatenate([[0], np.cumsum(counts)[:-1]])
    face_of = order // 3
    for v in range(n_vertices):
        c = counts[v]
        if c:
            table[v, :c] = face_of[starts[v]:starts[v] + c][::-1]
    return table


def sensor_tables(faces, vertex_ids):
    """
    Index tables used by the virtual-sensor code (reference virtual_sensors.py:47-75).
    :return: sub_faces (Fs,3) in ORIGINAL vertex numbering, sub_vertex_faces (M,deg) indexing `sub_faces` (-1 padded),
      helpers (M,) the helper vertex per sensor.
    """
    faces = np.asarray(faces, dtype=np.int64)
    n_vertices = int(faces.max()) + 1
    vf_full = vertex_faces_table(faces, n_vertices)
    v_ids = list(vertex_ids)
    rows = vf_full[v_ids]
    face_ids = np.unique(rows[rows != -1])
    sub_faces = faces[face_ids]
    vf_sub = vertex_faces_table(sub_faces, int(sub_faces.max()) + 1)[v_ids]
    helpers = []
    for v in v_ids:
        for cand in faces[vf_full[v, 0]]:  # first incident face, first vertex that is not v itself
            if cand != v:
                helpers.append(int(cand))
                break
    return sub_faces, vf_sub, np.asarray(helpers, dtype=np.int64)


# ----------------------------------------------------------------------------------------------------------------------
# SMPL-H body model (third-party `BodyModel`; PARITY UNPINNED, see module docstring).
# ----------------------------------------------------------------------------------------------------------------------
def rodrigues(rot_vecs, convention='smplx'):
    """
    Axis-angle -> matrix, R = I + sin(a) K + (1 - cos a) K^2 with K = hat(r / a).  (N,3) -> (N,3,3).  The two published
    conventions differ in how the angle a is guarded at r = 0 (SURVEY.md 8c "keep switchable"):
      'smplx'  a = ||r + 1e-8||                (smplx lbs.batch_rodrigues; what `BodyModel` is believed to use)
      'so3'    a = sqrt(clamp(||r||^2, 1e-4))  (pytorch3d so3_exponential_map, reference helpers/so3.py:116-121)
    """
    if convention == 'smplx':
        angle = torch.norm(rot_vecs + 1e-8, dim=1, keepdim=True)
    elif convention == 'so3':
        angle = torch.clamp((rot_vecs * rot_vecs).sum(1, keepdim=True), 1e-4).sqrt()
    else:
        raise ValueError('unknown Rodrigues convention {!r}'.format(convention))
    d = rot_vecs / angle
    cos = torch.cos(angle)[:, None]
    sin = torch.sin(angle)[:, None]
    rx, ry, rz = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    zeros = torch.zeros_like(rx)
    K = torch.cat([zeros, -rz, ry, rz, zeros, -rx, -ry, rx, zeros], dim=1).view(-1, 3, 3)
    ident = torch.eye(3, dtype=rot_vecs.dtype, device=rot_vecs.device)[None]
    return ident + sin * K + (1.0 - cos) * torch.bmm(K, K)


class BodyModelTensors(object):
    """The buffers `BodyModel` keeps, in its layouts: posedirs (459, V*3), J_regressor (52,V), weights (V,52)."""

    def __init__(self, model, num_betas=10, dtype=torch.float32, rodrigues_convention='smplx'):
        t = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float64)).to(dtype)
        self.dtype = dtype
        self.rodrigues_convention = rodrigues_convention
        self.v_template = t(model['v_template'])[None]  # (1,V,3)
        self.shapedirs = t(model['shapedirs'][:, :, :num_betas])  # (V,3,10)
        pd = np.asarray(model['posedirs'], dtype=np.float64)
        self.posedirs = t(pd.reshape(pd.shape[0] * 3, -1).T)  # (459, V*3)
        self.J_regressor = t(model['J_regressor'])  # (52,V)
        self.weights = t(model['weights'])  # (V,52)
        kt = np.asarray(model['kintree_table'])[0].astype(np.int64).copy()
        kt[0] = -1
        self.parents = kt.tolist()
        self.f = torch.from_numpy(np.asarray(model['f']).astype(np.int64))


def body_model_forward(bm, root_orient, pose_body, betas, pose_hand=None, trans=None):
    """
    SMPL-H evaluation as the reference calls it (reference smpl.py:121): returns (v (N,V,3), Jtr (N,52,3)).
    Follows the published lbs: shape blend, joint regression, Rodrigues, pose blend, rigid chain, skinning.
    """
    n = pose_body.shape[0]
    dt, dev = pose_body.dtype, pose_body.device
    if pose_hand is None:
        pose_hand = torch.zeros(n, 90, dtype=dt, device=dev)
    full_pose = torch.cat([root_orient, pose_body, pose_hand], dim=1)
    v_shaped = bm.v_template + torch.einsum('bl,mkl->bmk', betas, bm.shapedirs)
    J = torch.einsum('bik,ji->bjk', v_shaped, bm.J_regressor)
    n_j = J.shape[1]
    R = rodrigues(full_pose.reshape(-1, 3), getattr(bm, 'rodrigues_convention', 'smplx')).view(n, n_j, 3, 3)
    ident = torch.eye(3, dtype=dt, device=dev)
    pose_feature = (R[:, 1:] - ident).reshape(n, -1)
    v_posed = v_shaped + torch.matmul(pose_feature, bm.posedirs).view(n, -1, 3)

    parents = bm.parents
    rel = J.clone()
    rel[:, 1:] = J[:, 1:] - J[:, parents[1:]]
    T_local = torch.cat([torch.cat([R, rel[..., None]], dim=-1),
                         torch.tensor([0, 0, 0, 1], dtype=dt, device=dev).expand(n, n_j, 1, 4)], dim=2)
    chain = [T_local[:, 0]]
    for j in range(1, n_j):
        chain.append(torch.matmul(chain[parents[j]], T_local[:, j]))
    G = torch.stack(chain, dim=1)  # (N,52,4,4)
    Jtr = G[:, :, :3, 3]
    J_h = torch.cat([J, torch.zeros(n, n_j, 1, dtype=dt, device=dev)], dim=2)[..., None]
    A = G - torch.nn.functional.pad(torch.matmul(G, J_h), [3, 0])
    T = torch.matmul(bm.weights[None].expand(n, -1, -1), A.view(n, n_j, 16)).view(n, -1, 4, 4)
    v_h = torch.cat([v_posed, torch.ones(n, v_posed.shape[1], 1, dtype=dt, device=dev)], dim=2)
    v = torch.matmul(T, v_h[..., None])[:, :, :3, 0]
    if trans is not None:
        v = v + trans[:, None]
        Jtr = Jtr + trans[:, None]
    return v, Jtr


def smpl_fk(bm, poses_body, betas, poses_root=None, trans=None):
    """Wrapper semantics of the reference's SMPLLayer._fk (smpl.py:81-122): zero hands, zero trans, beta[:10]."""
    n = poses_body.shape[0]
    if poses_root is None:
        poses_root = torch.zeros(n, 3, dtype=poses_body.dtype)
    if betas.dim() == 1 or betas.shape[0] == 1:
        betas = betas.reshape(1, -1).repeat(n, 1)
    return body_model_forward(bm, poses_root, poses_body, betas[:, :10], None, trans)


# ----------------------------------------------------------------------------------------------------------------------
# Virtual sensors (reference virtual_sensors.py:16-38,77-96; utils.py:126-146).
# ----------------------------------------------------------------------------------------------------------------------
def vertex_normals_sub(vertices, sub_faces, sub_vertex_faces):
    """Un-normalised vertex normals at the sensor vertices: mean of incident (un-normalised) face normals."""
    vs = vertices[:, sub_faces]  # (N,Fs,3,3)
    fn = torch.cross(vs[:, :, 1] - vs[:, :, 0], vs[:, :, 2] - vs[:, :, 0], dim=-1)
    per = fn[:, sub_vertex_faces]  # (N,M,deg,3); -1 wraps to the last face and is zeroed next
    per = per * (sub_vertex_faces != -1).to(per.dtype)[None, :, :, None]
    deg = (sub_vertex_faces > -1).sum(dim=-1).to(per.dtype)
    return per.sum(dim=-2) / deg[None, :, None]


def sensor_frames(vertices, normals, vertex_ids, helper_ids):
    """Right-handed frame per sensor: columns (tangent towards helper re-orthogonalised, bitangent, unit normal)."""
    vs = vertices[:, vertex_ids]
    nh = normals / torch.norm(normals, dim=-1, keepdim=True)
    s = vertices[:, helper_ids] - vs
    s = s / torch.norm(s, dim=-1, keepdim=True)
    t = torch.cross(nh, s, dim=-1)
    t = t / torch.norm(t, dim=-1, keepdim=True)
    s2 = torch.cross(t, nh, dim=-1)
    s2 = s2 / torch.norm(s2, dim=-1, keepdim=True)
    return torch.stack([s2, t, nh], dim=-1)  # (N,M,3,3), columns


def virtual_pos_and_rot(vertices, vertex_ids, tables):
    sub_faces, vf_sub, helpers = tables
    sf = torch.from_numpy(sub_faces)
    vf = torch.from_numpy(vf_sub)
    normals = vertex_normals_sub(vertices, sf, vf)
    return vertices[:, list(vertex_ids)], sensor_frames(vertices, normals, list(vertex_ids), helpers.tolist()), normals


def estimated_markers(bm, tables, vertex_ids, poses, shapes, offset_r, offset_t):
    """Reference models.py:471-483."""
    v, joints = smpl_fk(bm, poses[:, 3:], shapes, poses[:, :3])
    pos, ori, _ = virtual_pos_and_rot(v, vertex_ids, tables)
    ori_c = torch.matmul(ori, offset_r)
    pos_c = pos + torch.matmul(ori, offset_t[..., None])[..., 0]
    return pos_c, ori_c, joints[:, :N_BODY]


# ----------------------------------------------------------------------------------------------------------------------
# Loss pieces (reference loss.py:23-41, utils.py:105-123).
# ----------------------------------------------------------------------------------------------------------------------
def mask_from_seq_lengths(seq_lengths, max_len=None):
    max_len = int(seq_lengths.max()) if max_len is None else max_len
    return torch.arange(max_len)[None, :] < seq_lengths[:, None]


def reconstruction_loss(gt, hat, seq_lengths=None, marker_mask=None):
    d = hat - gt
    per_frame = torch.sqrt((d * d).sum(-1)).sum(-1)  # (B,F)
    if marker_mask is not None:
        frame_ok = (marker_mask != 0).all(dim=-1)
        per_frame = per_frame * frame_ok
    if seq_lengths is not None:
        m = mask_from_seq_lengths(seq_lengths, per_frame.shape[1]).to(per_frame.dtype)
        per_frame = (per_frame * m).sum(-1) / seq_lengths.to(per_frame.dtype)
    return per_frame.mean()


def normal_mse(gt, hat, seq_lengths=None, marker_mask=None):
    """Squared error summed over joints, padded mean over frames, mean over the batch (reference loss.py:44-62)."""
    d = hat - gt
    per_frame = (d * d).sum(-1).sum(-1)
    if marker_mask is not None:
        per_frame = per_frame * (marker_mask != 0).all(dim=-1)
    if seq_lengths is not None:
        m = mask_from_seq_lengths(seq_lengths, per_frame.shape[1]).to(per_frame.dtype)
        per_frame = (per_frame * m).sum(-1) / seq_lengths.to(per_frame.dtype)
    return per_frame.mean()


def padded_l1(gt, hat, seq_lengths):
    """`padded_loss(gt, hat, nn.L1Loss(reduction='none'), seq_lengths)` (reference loss.py:13-20)."""
    per_frame = (gt - hat).abs().mean(-1)
    m = mask_from_seq_lengths(seq_lengths, per_frame.shape[1]).to(per_frame.dtype)
    return ((per_frame * m).sum(-1) / seq_lengths.to(per_frame.dtype)).mean()


def baseline_losses(out, poses, shapes, joints_gt, seq_lengths, marker_mask, fk_weight):
    """The loss of the two baselines (reference models.py:223-262, 326-366).  poses (B,F,66) root first, shapes (B,10),
    joints_gt (B,F,66).  :return: dict of the reference's loss values + 'total_loss' (tensors)."""
    B, F = poses.shape[0], poses.shape[1]
    r = lambda t: t.reshape(B, F, -1, 3)
    vals = {'pose': normal_mse(r(poses[:, :, 3:]), r(out['pose_hat']), seq_lengths, marker_mask),
            'root_pose': normal_mse(r(poses[:, :, :3]), r(out['root_ori_hat']), seq_lengths, marker_mask)}
    zero = torch.zeros((), dtype=poses.dtype)
    vals['shape'] = padded_l1(shapes.unsqueeze(1).repeat(1, F, 1), out['shape_hat'], seq_lengths) \
        if out['shape_hat'] is not None else zero
    vals['fk'] = reconstruction_loss(r(joints_gt), r(out['joints_hat']), seq_lengths, marker_mask) \
        if out['joints_hat'] is not None else zero
    vals['total_loss'] = vals['pose'] + vals['root_pose'] + vals['shape'] + fk_weight * vals['fk']
    return vals


# ----------------------------------------------------------------------------------------------------------------------
# Networks: the same torch.nn building blocks, addressed through a flat state_dict with the reference's key names
# (reference layers.py:13-77,80-157).
# ----------------------------------------------------------------------------------------------------------------------
def _bn_eval(y, sd, prefix, eps=1e-5):
    return (y - sd[prefix + 'running_mean']) / torch.sqrt(sd[prefix + 'running_var'] + eps) * sd[prefix + 'weight'] \
        + sd[prefix + 'bias']


def _prelu(y, a):
    return torch.where(y >= 0, y, a * y)


def mlp_forward(sd, prefix, x, num_layers=2, batch_norm=True, skip=False):
    """MLP in eval mode: Linear-BN-PReLU, `num_layers` blocks of 2x(Linear-BN-PReLU), Linear."""
    lin = lambda p, z: torch.addmm(sd[p + 'bias'], z, sd[p + 'weight'].t())
    y = lin(prefix + 'input_to_hidden.', x)
    if batch_norm:
        y = _bn_eval(y, sd, prefix + 'batch_norm.')
    y = _prelu(y, sd[prefix + 'activation_fn.weight'])
    for h in range(num_layers):
        z = y
        for k in range(2):
            step = 4 if batch_norm else 3
            base = prefix + 'hidden_layers.{}.layers.'.format(h)
            z = lin(base + '{}.'.format(k * step), z)
            if batch_norm:
                z = _bn_eval(z, sd, base + '{}.'.format(k * step + 1))
            z = _prelu(z, sd[base + '{}.weight'.format(k * step + (2 if batch_norm else 1))])
        y = y + z if skip else z
    return lin(prefix + 'hidden_to_output.', y)


def lstm_forward(sd, prefix, x, seq_lengths, state=None, num_layers=2, bidirectional=False):
    """
    Stacked (bi)directional LSTM over ragged sequences, written out explicitly (gate order i,f,g,o).
    Padded steps produce zero output and leave the state untouched (pack/pad semantics, reference layers.py:141-149);
    the reverse direction of row b starts at its last valid frame len_b-1.  States are indexed layer*dirs + direction.
    :return: y (B,F,dirs*H), (h_n, c_n) each (L*dirs,B,H)
    """
    B, F, _ = x.shape
    H = sd[prefix + 'weight_hh_l0'].shape[1]
    dirs = 2 if bidirectional else 1
    inp = x
    hs, cs = [], []
    for l in range(num_layers):
        per_dir = []
        for d in range(dirs):
            sfx = 'l%d' % l + ('_reverse' if d == 1 else '')
            w_ih, w_hh = sd[prefix + 'weight_ih_' + sfx], sd[prefix + 'weight_hh_' + sfx]
            b = sd[prefix + 'bias_ih_' + sfx] + sd[prefix + 'bias_hh_' + sfx]
            u = l * dirs + d
            h = torch.zeros(B, H, dtype=x.dtype) if state is None else state[0][u]
            c = torch.zeros(B, H, dtype=x.dtype) if state is None else state[1][u]
            outs = [None] * F
            for t in (range(F) if d == 0 else range(F - 1, -1, -1)):
                g = inp[:, t] @ w_ih.t() + h @ w_hh.t() + b
                i, f, gg, o = g[:, :H], g[:, H:2 * H], g[:, 2 * H:3 * H], g[:, 3 * H:]
                c_new = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
                h_new = torch.sigmoid(o) * torch.tanh(c_new)
                live = (t < seq_lengths)[:, None]
                c = torch.where(live, c_new, c)
                h = torch.where(live, h_new, h)
                outs[t] = torch.where(live, h_new, torch.zeros_like(h_new))
            per_dir.append(torch.stack(outs, dim=1))
            hs.append(h)
            cs.append(c)
        inp = torch.cat(per_dir, dim=-1)
    return inp, (torch.stack(hs), torch.stack(cs))


# ----------------------------------------------------------------------------------------------------------------------
# The two baselines (reference models.py:166-221 ResNet, 265-324 (Bi)RNN).
# ----------------------------------------------------------------------------------------------------------------------
def _baseline_inputs(inputs, n_markers, dt):
    pos, ori = inputs['marker_pos'].to(dt), inputs['marker_oris'].to(dt)
    B, F = pos.shape[0], pos.shape[1]
    m_pos, m_ori = pos.reshape(B, F, -1, 3), ori.reshape(B, F, -1, 9)
    if n_markers == 6:
        m_pos, m_ori = m_pos[:, :, S_CONFIG_6], m_ori[:, :, S_CONFIG_6]
    return torch.cat([m_pos.reshape(B, F, -1), m_ori.reshape(B, F, -1)], dim=-1)


def _baseline_heads(sd, bm, feats, estimate_shape, shape_avg, do_fk, skip):
    B, F = feats.shape[0], feats.shape[1]
    flat = feats.reshape(B * F, -1)
    pose = torch.addmm(sd['to_pose.bias'], flat, sd['to_pose.weight'].t()).reshape(B, F, -1)
    shape = joints = None
    if estimate_shape:
        shape = mlp_forward(sd, 'to_shape.', flat, 2, batch_norm=False, skip=skip).reshape(B, F, -1)
        if shape_avg:
            shape = shape.mean(dim=1, keepdim=True).repeat(1, F, 1)
    if do_fk:
        _, j = smpl_fk(bm, pose[:, :, 3:].reshape(B * F, -1), shape.reshape(B * F, -1),
                       poses_root=pose[:, :, :3].reshape(B * F, -1))
        joints = j[:, :22].reshape(B, F, -1)
    return {'pose_hat': pose[:, :, 3:], 'root_ori_hat': pose[:, :, :3], 'shape_hat': shape, 'joints_hat': joints}


def resnet_forward(sd, bm, inputs, n_markers=12, num_layers=3, estimate_shape=True, shape_avg=True, do_fk=True,
                   skip=False):
    """Frame-wise residual MLP: Linear, `num_layers` x relu(W x + b + x), heads (reference models.py:198-221)."""
    dt = sd['to_pose.weight'].dtype
    x = _baseline_inputs(inputs, n_markers, dt)
    B, F = x.shape[0], x.shape[1]
    y = torch.addmm(sd['from_input.bias'], x.reshape(B * F, -1), sd['from_input.weight'].t())
    for l in range(num_layers):
        y = torch.relu(torch.addmm(sd['blocks.%d.dense.bias' % l], y, sd['blocks.%d.dense.weight' % l].t()) + y)
    return _baseline_heads(sd, bm, y.reshape(B, F, -1), estimate_shape, shape_avg, do_fk, skip)


def simple_rnn_forward(sd, bm, inputs, n_markers=12, num_layers=2, bidirectional=True, learn_init_state=False,
                       state=None, estimate_shape=True, shape_avg=True, do_fk=True, skip=False):
    """(Bi)LSTM + heads (reference models.py:298-324). :return: (model_out, (h_n, c_n))"""
    dt = sd['to_pose.weight'].dtype
    x = _baseline_inputs(inputs, n_markers, dt)
    if learn_init_state:
        # reference layers.py:121-131 returns (c0, h0) and nn.LSTM reads the pair as (h_0, c_0): kept.
        H = sd['rnn.lstm.weight_hh_l0'].shape[1]
        mk = lambda n: torch.addmm(sd['rnn.to_init_state_%s.bias' % n], x[:, 0],
                                   sd['rnn.to_init_state_%s.weight' % n].t()).reshape(-1, num_layers, H).transpose(0, 1)
        state = (mk('c'), mk('h'))
    y, final = lstm_forward(sd, 'rnn.lstm.', x, inputs['seq_lengths'], state, num_layers, bidirectional)
    return _baseline_heads(sd, bm, y, estimate_shape, shape_avg, do_fk, skip), final


# ----------------------------------------------------------------------------------------------------------------------
# The LGD / IEF forward (reference models.py:485-632).
# ----------------------------------------------------------------------------------------------------------------------
S_CONFIG_6 = [0, 1, 2, 6, 7, 11]  # reference configuration.py:89


def ief_forward(sd, bm, tables, vertex_ids, inputs, n_markers=12, N=4, step_size=0.1, rnn_init=True,
                shape_avg=True, use_gradient=True, num_layers=2, batch_norm=True, skip=False, rnn_state=None,
                rnn_layers=2):
    """
    :param sd: state_dict (reference key names) of tensors in the working dtype.
    :param inputs: dict with marker_pos (B,F,36), marker_oris (B,F,108), offset_t (B,12,3), offset_r (B,12,3,3),
      marker_masks (B,F,12) or None, seq_lengths (B,) int64.
    :return: (model_out dict, trace dict). trace has the N+1 history entries and the per-iteration gradient features.
    """
    dt = sd['pose_net_iter.input_to_hidden.weight'].dtype
    pos = inputs['marker_pos'].to(dt)
    ori = inputs['marker_oris'].to(dt)
    B, F = pos.shape[0], pos.shape[1]
    T = B * F
    m_pos = pos.reshape(B, F, -1, 3)
    m_ori = ori.reshape(B, F, -1, 9)
    idx = list(range(12)) if n_markers == 12 else S_CONFIG_6
    if n_markers == 6:
        m_pos, m_ori = m_pos[:, :, idx], m_ori[:, :, idx]
    x_in = torch.cat([m_pos.reshape(B, F, -1), m_ori.reshape(B, F, -1)], dim=-1)
    x_flat = x_in.reshape(T, -1)
    seq_lengths = inputs['seq_lengths']
    masks = inputs.get('marker_masks')
    off_r = inputs['offset_r'].to(dt)[:, None].expand(B, F, 12, 3, 3).reshape(T, 12, 3, 3)
    off_t = inputs['offset_t'].to(dt)[:, None].expand(B, F, 12, 3).reshape(T, 12, 3)

    new_state = None
    if rnn_init:
        y, new_state = lstm_forward(sd, 'rnn.lstm.', x_in, seq_lengths, rnn_state, num_layers=rnn_layers)
        pose = (y @ sd['pose_net_init.weight'].t() + sd['pose_net_init.bias']).reshape(T, -1)
        shape = (y @ sd['shape_net_init.weight'].t() + sd['shape_net_init.bias']).reshape(T, -1)
    else:
        pose = mlp_forward(sd, 'pose_net_init.', x_flat, num_layers, batch_norm, skip)
        shape = mlp_forward(sd, 'shape_net_init.', x_flat, num_layers, batch_norm, skip)

    def window_mean(s):
        return s.reshape(B, F, -1).mean(dim=1, keepdim=True).expand(B, F, -1).reshape(T, -1)

    if shape_avg:
        shape = window_mean(shape)

    hist = {'pose': [], 'shape': [], 'joints': [], 'markers': [], 'markers_ori': [], 'g_pose': [], 'g_shape': []}

    def evaluate(p, s):
        return estimated_markers(bm, tables, vertex_ids, p, s, off_r, off_t)

    pose = pose.detach().requires_grad_(True)
    shape = shape.detach().requires_grad_(True)
    mp, mo, jt = evaluate(pose, shape)
    for i in range(N + 1):
        hist['pose'].append(pose.detach())
        hist['shape'].append(shape.detach())
        hist['joints'].append(jt.detach())
        hist['markers'].append(mp.detach())
        hist['markers_ori'].append(mo.detach())
        if i == N:
            break
        feats = [x_flat, pose.detach(), shape.detach()]
        if use_gradient:
            e = reconstruction_loss(m_pos, mp.reshape(B, F, 12, 3)[:, :, idx], seq_lengths, masks) + \
                reconstruction_loss(m_ori, mo.reshape(B, F, 12, 9)[:, :, idx], seq_lengths, masks)
            g_pose, g_shape = torch.autograd.grad(e, [pose, shape])
            g_pose, g_shape = g_pose * B * F, g_shape * B * F
            hist['g_pose'].append(g_pose)
            hist['g_shape'].append(g_shape)
            feats += [g_pose, g_shape]
        x = torch.cat(feats, dim=-1)
        d_pose = mlp_forward(sd, 'pose_net_iter.', x, num_layers, batch_norm, skip)
        d_shape = mlp_forward(sd, 'shape_net_iter.', x, num_layers, batch_norm, skip)
        if shape_avg:
            d_shape = window_mean(d_shape)
        pose = (pose.detach() + d_pose * step_size).detach().requires_grad_(True)
        shape = (shape.detach() + d_shape * step_size).detach().requires_grad_(True)
        mp, mo, jt = evaluate(pose, shape)

    pose_f = pose.detach().reshape(B, F, -1)
    out = {'pose_hat': pose_f[:, :, 3:], 'root_ori_hat': pose_f[:, :, :3],
           'shape_hat': shape.detach().reshape(B, F, -1), 'joints_hat': jt.detach().reshape(B, F, -1)}
    hist['rnn_state'] = new_state
    return out, hist
