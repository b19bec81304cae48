"""
SMPL-H layer (mirror of reference empose/bodymodels/smpl.py:24-165) backed by the HIP full-mesh kernels.

`SMPLLayer(...)(poses_body, betas, poses_root=None, trans=None, normalize_root=False, window_size=None)` returns
`(vertices (N,V,3), joints (N,52,3))` exactly as the reference (`body.v`, `body.Jtr`, smpl.py:121-122): the 30 hand
joints have zero pose (smpl.py:99) and ride rigidly on the wrists.  Differences to the reference, on purpose:
  * `normalize_root=True` is not implemented (never used on this path, reference smpl.py:112-119).
  * the LGD loop itself never calls this layer: `IterativeErrorFeedback` evaluates only the sensor sub-mesh.
  * `rodrigues_convention` ('smplx' | 'so3') selects how the axis-angle map guards the angle at zero; the fork that
    holds the reference's arithmetic is not available, so the choice is explicit (include/empose_hip.h).

Buffers live under `self.bm` with the names of the third-party `BodyModel` (`f, v_template, shapedirs, posedirs,
J_regressor, weights`) so that `state_dict` keys of released checkpoints (`smpl.bm.*`) match.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from em_pose_amd import _lib
from em_pose_amd.bodymodels import tables as TB
from em_pose_amd.helpers.configuration import CONSTANTS as C


def load_model_npz(path_or_dict):
    """Read the arrays of an SMPL-H `model.npz` (or take an equivalent dict)."""
    src = path_or_dict if isinstance(path_or_dict, dict) else np.load(path_or_dict, allow_pickle=True)
    keys = ('v_template', 'f', 'shapedirs', 'posedirs', 'J_regressor', 'weights', 'kintree_table')
    model = {k: np.asarray(src[k]) for k in keys}
    if model['posedirs'].ndim != 3 or model['posedirs'].shape[2] != 459:
        raise ValueError('expected SMPL-H posedirs of shape (V,3,459), got {}'.format(model['posedirs'].shape))
    if model['J_regressor'].shape[0] != 52:
        raise ValueError('expected a 52-joint SMPL-H model')
    return model


class _BodyModelBuffers(nn.Module):
    """Holds the model arrays under the third-party module's buffer names and layouts."""

    def __init__(self, model, num_betas):
        super(_BodyModelBuffers, self).__init__()
        t = lambda a: torch.from_numpy(np.array(a, dtype=np.float32))     # own memory: `model` must not alias buffers
        self.register_buffer('f', torch.from_numpy(np.array(model['f'], dtype=np.int64)))
        self.register_buffer('v_template', t(model['v_template'])[None])
        self.register_buffer('shapedirs', t(model['shapedirs'][:, :, :num_betas]))
        pd = np.asarray(model['posedirs'], dtype=np.float32)
        self.register_buffer('posedirs', t(pd.reshape(pd.shape[0] * 3, -1).T).contiguous())
        self.register_buffer('J_regressor', t(model['J_regressor']))
        self.register_buffer('weights', t(model['weights']))
        # The fork registers its default pose/shape as nn.Parameters (169 values, reference README.md:228).
        self.trans = nn.Parameter(torch.zeros(1, 3))
        self.root_orient = nn.Parameter(torch.zeros(1, 3))
        self.pose_body = nn.Parameter(torch.zeros(1, 63))
        self.pose_hand = nn.Parameter(torch.zeros(1, 90))
        self.betas = nn.Parameter(torch.zeros(1, num_betas))


class SMPLLayer(nn.Module):
    def __init__(self, smpl_path, device=None, vposer_path=None, rodrigues_convention='smplx', arithmetic='f32'):
        super(SMPLLayer, self).__init__()
        if vposer_path is not None:
            raise NotImplementedError('VPoser is not part of the LGD path')
        if rodrigues_convention not in _lib.RODRIGUES:
            raise ValueError('rodrigues_convention must be one of {}'.format(sorted(_lib.RODRIGUES)))
        self.rodrigues_convention = rodrigues_convention
        if arithmetic not in ('f32', 'bf16x3'):
            raise ValueError("arithmetic must be 'f32' (exact fp32 matrix cores, the default) or 'bf16x3'")
        self.arithmetic = arithmetic   # 'bf16x3': blend-shape contraction in split bf16 (explicit opt-in, not the reference's)
        self.num_betas = C.N_SHAPE_PARAMS
        self.model = load_model_npz(smpl_path)
        self.bm = _BodyModelBuffers(self.model, self.num_betas)
        self._faces = None
        self._vertex_faces = None
        self._mesh = None  # (handle, device index)
        self.tables_version = 0   # bumped when the arrays change: holders of tables derived from `self.model` rebuild
        # A released `model.pth` carries the body model as `smpl.bm.*` buffers and the reference's network computes with
        # THOSE after `load_state_dict` (reference eval/helpers.py:131-137), whatever `model.npz` was read at construction.
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._adopt_buffers())

    def _adopt_buffers(self):
        """Make the kernel tables follow the `bm` buffers (after load_state_dict).  No-op when they already agree."""
        bm, m, nb = self.bm, self.model, self.num_betas
        V = bm.v_template.shape[1]
        new = {'v_template': bm.v_template[0], 'shapedirs': bm.shapedirs, 'J_regressor': bm.J_regressor,
               'weights': bm.weights, 'posedirs': bm.posedirs.t().reshape(V, 3, -1), 'f': bm.f}
        new = {k: v.detach().cpu().numpy() for k, v in new.items()}
        old = {k: (np.asarray(m[k])[:, :, :nb] if k == 'shapedirs' else np.asarray(m[k])) for k in new}
        if all(old[k].shape == new[k].shape and np.array_equal(old[k].astype(new[k].dtype), new[k]) for k in new):
            return
        model = dict(m)
        model.update(new)
        self.model = model
        self._faces = self._vertex_faces = None
        self._release()
        self.tables_version += 1

    # -- topology ----------------------------------------------------------------------------------------------
    @property
    def n_vertices(self):
        return self.model['v_template'].shape[0]

    @property
    def faces(self):
        if self._faces is None:
            self._faces = self.bm.f.to(dtype=torch.int32)
        return self._faces

    def vertex_faces(self, n_vertices):
        if self._vertex_faces is None:
            vf = TB.vertex_faces_table(self.model['f'], n_vertices)
            self._vertex_faces = torch.from_numpy(vf).to(dtype=torch.long, device=self.bm.f.device)
        return self._vertex_faces

    def vertex_normals(self, vertices, output_vertex_ids=None):
        """Un-normalised vertex normals of posed meshes (N, V, 3) -> (N, V, 3), or (N, len(ids), 3) for the given vertex
        ids (reference smpl.py:69-79)."""
        from em_pose_amd.data.virtual_sensors import VirtualMarkerHelper
        if getattr(self, '_normal_helper', None) is None:
            self._normal_helper = VirtualMarkerHelper(self)
        ids = list(range(vertices.shape[1])) if output_vertex_ids is None else [int(i) for i in output_vertex_ids]
        return self._normal_helper.get_vertex_normals(vertices, ids)

    # -- HIP full-mesh evaluation -----------------------------------------------------------------------------
    @property
    def n_joints(self):
        return int(np.asarray(self.model['J_regressor']).shape[0])

    def _mesh_handle(self, device):
        key = (device.index, self.rodrigues_convention, self.arithmetic)
        if self._mesh is not None and self._mesh[1] == key:
            return self._mesh[0]
        self._release()
        tab = TB.build_full_mesh_tables(self.model, self.num_betas)
        desc = _lib.MeshDesc()
        desc.n_vertices, desc.j_off, desc.ncp, desc.kb = tab['n_vertices'], tab['j_off'], tab['ncp'], tab['kb']
        desc.wc, desc.skin_idx = _lib.fptr(tab['wc']), _lib.iptr(tab['skin_idx'])
        desc.skin_w, desc.parents = _lib.fptr(tab['skin_w']), _lib.iptr(tab['parents'])
        desc.n_joints, desc.rodrigues = tab['n_joints'], _lib.RODRIGUES[self.rodrigues_convention]
        desc.with_bf16x3 = int(self.arithmetic == 'bf16x3')
        handle = _lib.C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(_lib.lib().empose_mesh_create(_lib.C.byref(desc), _lib.C.byref(handle)))
        self._mesh = (handle, key)
        return handle

    def _release(self):
        if self._mesh is not None:
            _lib.lib().empose_mesh_destroy(self._mesh[0])
            self._mesh = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _fk(self, poses_body, betas, poses_root=None, trans=None, normalize_root=False):
        assert poses_body.shape[1] >= C.N_JOINTS * 3
        if normalize_root:
            raise NotImplementedError('normalize_root is not available on the HIP path')
        if not poses_body.is_cuda:
            raise _lib.EmposeError('SMPLLayer needs GPU tensors; there is no CPU fallback')
        n, dev = poses_body.shape[0], poses_body.device
        if poses_root is None:
            poses_root = torch.zeros(n, 3, dtype=torch.float32, device=dev)
        if betas.dim() == 1 or betas.shape[0] == 1:
            betas = betas.reshape(1, -1).repeat(n, 1)
        betas = betas[:, :self.num_betas].contiguous().float()
        poses = torch.cat([poses_root.float(), poses_body[:, :C.N_JOINTS * 3].float()], dim=1).contiguous()
        trans = trans.contiguous().float() if trans is not None else None
        lib = _lib.lib()
        with torch.cuda.device(dev):
            handle = self._mesh_handle(dev)
            vertices = torch.empty(n, self.n_vertices, 3, dtype=torch.float32, device=dev)
            joints = torch.empty(n, self.n_joints, 3, dtype=torch.float32, device=dev)
            ws_bytes = lib.empose_mesh_workspace_bytes(handle, n)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            fwd = lib.empose_mesh_vertices_fwd_bf16x3 if self.arithmetic == 'bf16x3' else lib.empose_mesh_vertices_fwd
            _lib.check(fwd(handle, n, _lib.dptr(poses), _lib.dptr(betas), _lib.dptr(trans),
                                                    _lib.dptr(vertices), _lib.dptr(joints), _lib.dptr(ws), ws_bytes,
                                                    _lib.current_stream()))
        return vertices, joints

    def fk_joints(self, poses_body, betas, poses_root=None, trans=None):
        """The 22 posed body joints only (no vertices): forward kinematics for the metrics, (N,22,3) contiguous
        (= `fk(...)[1][:, :22]`, what reference eval/metrics.py:223-228 keeps)."""
        if not poses_body.is_cuda:
            raise _lib.EmposeError('SMPLLayer needs GPU tensors; there is no CPU fallback')
        n, dev = poses_body.shape[0], poses_body.device
        if poses_root is None:
            poses_root = torch.zeros(n, 3, dtype=torch.float32, device=dev)
        if betas.dim() == 1 or betas.shape[0] == 1:
            betas = betas.reshape(1, -1).repeat(n, 1)
        betas = betas[:, :self.num_betas].contiguous().float()
        poses = torch.cat([poses_root.float(), poses_body[:, :C.N_JOINTS * 3].float()], dim=1).contiguous()
        trans = trans.contiguous().float() if trans is not None else None
        lib = _lib.lib()
        with torch.cuda.device(dev):
            handle = self._mesh_handle(dev)
            joints = torch.empty(n, self.n_joints, 3, dtype=torch.float32, device=dev)
            ws_bytes = lib.empose_mesh_workspace_bytes(handle, n)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            _lib.check(lib.empose_mesh_joints_fwd(handle, n, _lib.dptr(poses), _lib.dptr(betas), _lib.dptr(trans),
                                                  _lib.dptr(joints), _lib.dptr(ws), ws_bytes, _lib.current_stream()))
        return joints[:, :C.N_JOINTS + 1].contiguous()

    def fk(self, poses_body, betas, poses_root=None, trans=None, normalize_root=False, window_size=None):
        # The reference slices long inputs into windows to bound memory (smpl.py:124-144); the HIP entry point already
        # processes slabs of 2048 frames internally, so `window_size` only has to be accepted.
        if window_size is not None and normalize_root:
            raise ValueError('Are you sure you want to use root normalization with windowed evaluation?')
        return self._fk(poses_body, betas, poses_root, trans, normalize_root)

    def forward(self, *args, **kwargs):
        return self.fk(*args, **kwargs)


def create_default_smpl_model(device=None, vposer_path=None):
    """Loads `$SMPL_MODELS/smplh_amass/neutral/model.npz` (reference smpl.py:24-28)."""
    device = C.DEVICE if device is None else device
    layer = SMPLLayer(os.path.join(C.SMPL_MODELS_DIR, 'smplh_amass/neutral/model.npz'), vposer_path=vposer_path)
    return layer.to(device=device, dtype=torch.float32)
