"""
Model-constant preprocessing for the HIP kernels (host side, runs once at model load, NumPy float64 -> float32).

What the LGD loop needs from SMPL-H is tiny compared to what the reference evaluates: only the
12 sensor vertices, the vertices of their incident faces and one helper vertex each (~84 of 6890 vertices) are ever
read (reference virtual_sensors.py:61-96), only the 22 body joints carry a rotation (hands are zero,
reference smpl.py:99) and only joints[:22] are returned (reference models.py:481).  This module derives the packed
constants for that restricted evaluation:

* `wc` / `wct`  -- one dense matrix that maps the per-frame feature vector
                   feat = [vec(R_1 - I), ..., vec(R_21 - I) (189) | beta (10) | 1]            (K = 200)
                   to     out  = [v_posed of the needed vertices (NV*3) | rest joints J (66)]  (N = NCP)
                   i.e. template + shape blend-shapes + pose blend-shapes + joint regression in ONE fp32 GEMM.
* skinning weights folded from 52 to 22 joints (hand joints are rigidly attached to the wrists, their relative
  transform equals the wrist's), stored per vertex (forward) and per bone (CSR, backward gather).
* sensor patches in local (needed-vertex) numbering: centre, helper, incident faces.
* kinematic tree walks: root->joint paths (forward) and subtree lists (backward).

The vertex->faces order and the helper-vertex choice follow trimesh's `vertex_faces` as the reference uses it
(reference smpl.py:58-67, virtual_sensors.py:47-59); callers that have the exact tables of a trained model can pass
them in as data (`helper_ids`) instead of having them re-derived.
"""
import numpy as np

N_BODY = 22
K_FEAT = 200  # 189 pose-feature columns + 10 betas + 1


def vertex_faces_table(faces, n_vertices):
    """
    (V, max_degree) table of incident face ids per vertex, -1 padded, in the order of trimesh==3.9.32's
    `Trimesh.vertex_faces` (reference smpl.py:58-67, virtual_sensors.py:47-75): DESCENDING face id.  trimesh fills the
    rows from `faces_sparse.dot(identity).nonzero()[1]` (`geometry.vertex_face_indices`); scipy's CSR product emits the
    columns of a row in reverse insertion order, and trimesh's own slow-loop fallback reverses explicitly to match.  The
    order matters: `vertex_faces[v, 0]` picks the helper vertex of a sensor frame (reference virtual_sensors.py:55),
    i.e. the in-plane axes every learned offset lives in.  tests/test_host_logic.py pins this table to the scipy
    expression.
    """
    faces = np.asarray(faces, dtype=np.int64)
    flat = faces.reshape(-1)
    counts = np.bincount(flat, minlength=n_vertices)
    order = np.argsort(flat, kind='stable') // 3   # ascending face ids per vertex
    starts = np.concatenate([[0], np.cumsum(counts)])
    table = np.full((n_vertices, max(int(counts.max()), 1)), -1, dtype=np.int64)
    for v in range(n_vertices):
        table[v, :counts[v]] = order[starts[v]:starts[v + 1]][::-1]
    return table


def sensor_topology(faces, vertex_ids):
    """
    :return: (sub_faces (Fs,3) original numbering, sub_vertex_faces (M,deg) into sub_faces, helpers (M,))
    """
    faces = np.asarray(faces, dtype=np.int64)
    n_vertices = int(faces.max()) + 1
    vf = vertex_faces_table(faces, n_vertices)
    rows = vf[list(vertex_ids)]
    face_ids = np.unique(rows[rows != -1])
    sub_faces = faces[face_ids]
    vf_sub = vertex_faces_table(sub_faces, int(sub_faces.max()) + 1)[list(vertex_ids)]
    helpers = []
    for v in vertex_ids:
        first = faces[vf[v, 0]]
        helpers.append(int([c for c in first if c != v][0]))
    return sub_faces, vf_sub, np.asarray(helpers, dtype=np.int64)


def _round_up(x, m):
    return (x + m - 1) // m * m


def fold_parents(kintree_parents):
    """Map every joint to the body joint (<22) whose rigid transform it shares when hand rotations are zero."""
    parents = list(kintree_parents)
    fold = list(range(len(parents)))
    for j in range(N_BODY, len(parents)):
        a = j
        while a >= N_BODY:
            a = parents[a]
        fold[j] = a
    return fold


def fold_weights(weights, kintree_parents):
    """(V,52) -> (V,22): add the hand-joint columns onto their wrist."""
    w = np.asarray(weights, dtype=np.float64)
    out = np.zeros((w.shape[0], N_BODY))
    for j, a in enumerate(fold_parents(kintree_parents)):
        out[:, a] += w[:, j]
    return out


def _sparsify(w22, kb=None):
    nnz = (w22 != 0).sum(axis=1)
    kb = int(max(nnz.max(), 1)) if kb is None else kb
    idx = np.zeros((w22.shape[0], kb), dtype=np.int32)
    val = np.zeros((w22.shape[0], kb), dtype=np.float64)
    for s in range(w22.shape[0]):
        nz = np.nonzero(w22[s])[0]
        idx[s, :len(nz)] = nz
        val[s, :len(nz)] = w22[s, nz]
    return idx, val, kb


def tree_walks(parents):
    """root->j paths (inclusive) and subtree member lists, both CSR."""
    n = len(parents)
    path_ptr, path = [0], []
    for j in range(n):
        p, a = [], j
        while a >= 0:
            p.append(a)
            a = parents[a]
        path.extend(p[::-1])
        path_ptr.append(len(path))
    sub_ptr, sub = [0], []
    for j in range(n):
        members = []
        for d in range(n):
            a = d
            while a >= 0 and a != j:
                a = parents[a]
            if a == j:
                members.append(d)
        sub.extend(members)
        sub_ptr.append(len(sub))
    return (np.asarray(path_ptr, np.int32), np.asarray(path, np.int32),
            np.asarray(sub_ptr, np.int32), np.asarray(sub, np.int32))


def model_parents(model):
    kt = np.asarray(model['kintree_table'])[0].astype(np.int64).copy()
    kt[0] = -1
    return kt.tolist()


def pose_rows(model, num_betas=10):
    """
    Per output coordinate (V*3 rows) the K_FEAT coefficients [posedirs live columns | shapedirs | template].
    posedirs in the asset are (V,3,459) with the 459 axis = (joint-1)*9 + 3*r + c over joints 1..51; only joints
    1..21 can be non-identity on this path, i.e. the first 189 columns.
    """
    V = model['v_template'].shape[0]
    pd = np.asarray(model['posedirs'], dtype=np.float64).reshape(V * 3, -1)[:, :189]
    sd = np.asarray(model['shapedirs'], dtype=np.float64)[:, :, :num_betas].reshape(V * 3, num_betas)
    vt = np.asarray(model['v_template'], dtype=np.float64).reshape(V * 3, 1)
    return np.concatenate([pd, sd, vt], axis=1)  # (V*3, 200)


def joint_rows(model, num_betas=10, n_joints=N_BODY):
    """Rows for the first `n_joints` rest joints: J = J_reg @ (v_template + shapedirs beta)  ->
    [0 (189) | J_S (10) | J_t (1)]."""
    Jr = np.asarray(model['J_regressor'], dtype=np.float64)[:n_joints]
    sd = np.asarray(model['shapedirs'], dtype=np.float64)[:, :, :num_betas]
    vt = np.asarray(model['v_template'], dtype=np.float64)
    J_t = Jr @ vt  # (22,3)
    J_S = np.einsum('jv,vkl->jkl', Jr, sd)  # (22,3,10)
    rows = np.zeros((n_joints * 3, K_FEAT))
    rows[:, 189:199] = J_S.reshape(n_joints * 3, num_betas)
    rows[:, 199] = J_t.reshape(-1)
    return rows


def build_lgd_tables(model, vertex_ids, helper_ids=None, num_betas=10, dtype=np.float32):
    """All constants of the sub-mesh evaluation, as a dict of C-contiguous float32 / int32 arrays."""
    assert num_betas == 10
    faces = np.asarray(model['f'], dtype=np.int64)
    vertex_ids = [int(v) for v in vertex_ids]
    M = len(vertex_ids)
    parents52 = model_parents(model)
    parents = parents52[:N_BODY]
    assert all(p < j for j, p in enumerate(parents)), 'joints must be topologically ordered'

    sub_faces, vf_sub, helpers = sensor_topology(faces, vertex_ids)
    if helper_ids is not None:
        helpers = np.asarray(helper_ids, dtype=np.int64)

    # Needed vertices, local numbering: sensor by sensor (centre, helper, ring) without repeats.
    needed, local = [], {}

    def loc(v):
        v = int(v)
        if v not in local:
            local[v] = len(needed)
            needed.append(v)
        return local[v]

    deg = (vf_sub >= 0).sum(axis=1).astype(np.int32)
    max_deg = int(deg.max())
    s_center = np.zeros(M, np.int32)
    s_helper = np.zeros(M, np.int32)
    s_faces = np.zeros((M, max_deg, 3), np.int32)
    for m, v in enumerate(vertex_ids):
        s_center[m] = loc(v)
        s_helper[m] = loc(helpers[m])
        for k in range(deg[m]):
            for c in range(3):
                s_faces[m, k, c] = loc(sub_faces[vf_sub[m, k], c])
    needed = np.asarray(needed, dtype=np.int64)
    NV = len(needed)
    NV3 = NV * 3
    j_off = _round_up(NV3, 4)
    ncp = _round_up(j_off + N_BODY * 3, 4)

    rows_all = pose_rows(model, num_betas)
    wc = np.zeros((ncp, K_FEAT))
    sel = (needed[:, None] * 3 + np.arange(3)[None, :]).reshape(-1)
    wc[:NV3] = rows_all[sel]
    wc[j_off:j_off + N_BODY * 3] = joint_rows(model, num_betas)

    w22 = fold_weights(np.asarray(model['weights'])[needed], parents52)
    skin_idx, skin_w, kb = _sparsify(w22)
    bone_ptr, bone_vert, bone_w = [0], [], []
    for b in range(N_BODY):
        for s in range(NV):
            if w22[s, b] != 0:
                bone_vert.append(s)
                bone_w.append(w22[s, b])
        bone_ptr.append(len(bone_vert))

    path_ptr, path, sub_ptr, sub = tree_walks(parents)
    f32 = lambda a: np.ascontiguousarray(a, dtype=dtype)
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    return {
        'n_sensors': M, 'nv': NV, 'j_off': j_off, 'ncp': ncp, 'kb': kb, 'max_deg': max_deg,
        'needed': i32(needed), 'parents': i32(parents),
        'wc': f32(wc), 'wct': f32(wc.T),
        'skin_idx': i32(skin_idx), 'skin_w': f32(skin_w),
        'bone_ptr': i32(bone_ptr), 'bone_vert': i32(bone_vert), 'bone_w': f32(bone_w),
        's_center': i32(s_center), 's_helper': i32(s_helper), 's_deg': i32(deg), 's_faces': i32(s_faces),
        'path_ptr': path_ptr, 'path': path, 'sub_ptr': sub_ptr, 'sub': sub,
    }


def build_full_mesh_tables(model, num_betas=10, dtype=np.float32, n_joints=None):
    """Constants for the full-mesh vertex kernel (final vertices / ground-truth preprocessing).  `n_joints`: how many
    posed joints the kernel returns -- all of the model's (52 for SMPL-H, what the reference's `body.Jtr` holds,
    smpl.py:121-122) by default; the joints past the 22 body joints have zero pose and only need their rest position
    and parent."""
    parents52 = model_parents(model)
    n_joints = len(parents52) if n_joints is None else int(n_joints)
    assert N_BODY <= n_joints <= len(parents52)
    assert all(p < j for j, p in enumerate(parents52[:n_joints])), 'joints must be topologically ordered'
    V = model['v_template'].shape[0]
    w22 = fold_weights(model['weights'], parents52)
    skin_idx, skin_w, kb = _sparsify(w22)
    rows = np.concatenate([pose_rows(model, num_betas), joint_rows(model, num_betas, n_joints)], axis=0)
    n_rows = _round_up(rows.shape[0], 4)
    w = np.zeros((n_rows, K_FEAT))
    w[:rows.shape[0]] = rows
    return {'n_vertices': V, 'j_off': V * 3, 'ncp': n_rows, 'kb': kb, 'n_joints': n_joints,
            'wc': np.ascontiguousarray(w, dtype=dtype),
            'skin_idx': np.ascontiguousarray(skin_idx, dtype=np.int32),
            'skin_w': np.ascontiguousarray(skin_w, dtype=dtype),
            'parents': np.ascontiguousarray(parents52[:n_joints], dtype=np.int32)}
