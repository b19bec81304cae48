"""
Virtual sensors on a posed mesh (mirror of reference empose/data/virtual_sensors.py:41-96), HIP-backed.

`VirtualMarkerHelper(smpl_model).get_virtual_pos_and_rot(vertices (N,V,3), vertex_ids)` returns
`(positions (N,M,3), orientations (N,M,3,3), un-normalised vertex normals (N,M,3))` like the reference.  The index
tables (faces incident to the sensor vertices, helper vertex = first other vertex of the first incident face) are
derived once per `vertex_ids` tuple and cached on the device, as the reference caches them with `lru_cache`.
Inside the LGD loop this class is not used: `IterativeErrorFeedback` evaluates the sensors straight from the sub-mesh.
"""
import numpy as np
import torch

from em_pose_amd import _lib
from em_pose_amd.bodymodels import tables as TB


class VirtualMarkerHelper(object):
    def __init__(self, smpl_model):
        self.smpl_model = smpl_model
        self._cache = {}

    def _topology(self, vertex_ids):
        faces = np.asarray(self.smpl_model.model['f'], dtype=np.int64)
        sub_faces, vf_sub, helpers = TB.sensor_topology(faces, list(vertex_ids))
        return sub_faces, vf_sub, helpers

    def get_vertex_helpers(self, vertex_ids):
        return self._topology(tuple(vertex_ids))[2].tolist()

    def get_sub_faces(self, vertex_ids):
        sub_faces, vf_sub, _ = self._topology(tuple(vertex_ids))
        return torch.from_numpy(sub_faces).long(), torch.from_numpy(vf_sub).long()

    def _tables(self, vertex_ids, device):
        key = (tuple(vertex_ids), str(device))
        if key not in self._cache:
            sub_faces, vf_sub, helpers = self._topology(tuple(vertex_ids))
            deg = (vf_sub >= 0).sum(axis=1).astype(np.int32)
            max_deg = int(deg.max())
            faces = np.zeros((len(vertex_ids), max_deg, 3), dtype=np.int32)
            for m in range(len(vertex_ids)):
                faces[m, :deg[m]] = sub_faces[vf_sub[m, :deg[m]]]
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(device)
            self._cache[key] = (t(np.asarray(vertex_ids)), t(helpers), t(deg), t(faces), max_deg)
        return self._cache[key]

    def get_virtual_pos_and_rot(self, vertices, vertex_ids):
        if not vertices.is_cuda:
            raise _lib.EmposeError('VirtualMarkerHelper needs GPU tensors; there is no CPU fallback')
        v = vertices.contiguous().float()
        n, nv = v.shape[0], v.shape[1]
        center, helper, deg, faces, max_deg = self._tables(vertex_ids, v.device)
        m = len(vertex_ids)
        pos = torch.empty(n, m, 3, dtype=torch.float32, device=v.device)
        ori = torch.empty(n, m, 3, 3, dtype=torch.float32, device=v.device)
        nor = torch.empty(n, m, 3, dtype=torch.float32, device=v.device)
        with torch.cuda.device(v.device):
            _lib.check(_lib.lib().empose_virtual_sensors_fwd(n, nv, _lib.dptr(v), m, max_deg, _lib.dptr(center),
                                                             _lib.dptr(helper), _lib.dptr(deg), _lib.dptr(faces),
                                                             _lib.dptr(pos), _lib.dptr(ori), _lib.dptr(nor),
                                                             _lib.current_stream()))
        return pos, ori, nor

    def get_vertex_normals(self, vertices, vertex_ids):
        """Un-normalised vertex normals (mean of the incident faces' (v1-v0)x(v2-v0)) at `vertex_ids`, (N, M, 3)
        (reference virtual_sensors.py:77-83)."""
        return self.get_virtual_pos_and_rot(vertices, vertex_ids)[2]
