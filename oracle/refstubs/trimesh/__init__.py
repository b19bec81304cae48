"""Stand-in for trimesh==3.9.32 (reference requirements.txt:39): only `Trimesh(...).vertex_faces`."""
import numpy as np

from oracle.torch_ref import vertex_faces_table


class Trimesh(object):
    def __init__(self, vertices=None, faces=None, process=False, **kwargs):
        self.vertices = np.asarray(vertices)
        self.faces = np.asarray(faces, dtype=np.int64)

    @property
    def vertex_faces(self):
        return vertex_faces_table(self.faces, self.vertices.shape[0])
