"""
Stand-in for human_body_prior.body_model.body_model.BodyModel (fork @821a0e7, reference requirements.txt:9).
PARITY UNPINNED: the arithmetic is oracle.torch_ref.body_model_forward. Buffer names follow the upstream project
(`f, v_template, shapedirs, posedirs, J_regressor, weights`) so state_dict keys look like the real ones.
"""
import numpy as np
import torch
import torch.nn as nn

from oracle.torch_ref import BodyModelTensors, body_model_forward


class _Out(object):
    pass


class BodyModel(nn.Module):
    def __init__(self, bm_path, num_betas=10, dtype=torch.float64, **kwargs):
        super(BodyModel, self).__init__()
        model = bm_path if isinstance(bm_path, dict) else dict(np.load(bm_path))
        t = BodyModelTensors(model, num_betas=num_betas, dtype=dtype)
        self.register_buffer('f', t.f)
        self.register_buffer('v_template', t.v_template)
        self.register_buffer('shapedirs', t.shapedirs)
        self.register_buffer('posedirs', t.posedirs)
        self.register_buffer('J_regressor', t.J_regressor)
        self.register_buffer('weights', t.weights)
        self.parents = t.parents

    def forward(self, root_orient=None, pose_body=None, betas=None, pose_hand=None, trans=None, **kwargs):
        v, jtr = body_model_forward(self, root_orient, pose_body, betas, pose_hand, trans)
        out = _Out()
        out.v, out.Jtr, out.f = v, jtr, self.f
        return out
