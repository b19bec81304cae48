"""
Stand-in for trimesh==3.9.32 (reference requirements.txt:39): only `Trimesh(...).vertex_faces`.

`vertex_faces` restates the published algorithm of trimesh's `geometry.vertex_face_indices` [upstream-knowledge]: a
(V x F) boolean COO incidence matrix (`geometry.index_sparse`), multiplied by an integer identity, and the column
indices of the product's non-zeros packed row by row into a -1 padded table.  The row order is whatever scipy's sparse
product yields (descending face id), which is why this stand-in evaluates the scipy expression instead of sorting:
`oracle.torch_ref.vertex_faces_table` / `em_pose_amd.bodymodels.tables.vertex_faces_table` state that order directly
and are checked against this module in tests/test_host_logic.py.
"""
import numpy as np
import scipy.sparse


def _faces_sparse(n_vertices, faces):
    row = faces.reshape(-1)
    col = np.tile(np.arange(len(faces)).reshape((-1, 1)), (1, faces.shape[1])).reshape(-1)
    data = np.ones(len(col), dtype=bool)
    return scipy.sparse.coo_matrix((data, (row, col)), shape=(n_vertices, len(faces)), dtype=data.dtype)


def vertex_face_indices(vertex_count, faces):
    counts = np.bincount(faces.flatten(), minlength=vertex_count)
    starts = np.append(0, np.cumsum(counts)[:-1])
    pack = np.arange(counts.max()) + starts[:, None]
    padded = -(pack >= (starts + counts)[:, None]).astype(np.int64)
    identity = scipy.sparse.identity(len(faces), dtype=int)
    sorted_faces = _faces_sparse(vertex_count, faces).dot(identity).nonzero()[1]
    padded[padded == 0] = sorted_faces
    return padded


class Trimesh(object):
    def __init__(self, vertices=None, faces=None, process=False, **kwargs):
        self.vertices = np.asarray(vertices)
        self.faces = np.asarray(faces, dtype=np.int64)

    @property
    def vertex_faces(self):
        return vertex_face_indices(self.vertices.shape[0], self.faces)
