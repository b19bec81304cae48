"""CPU-side checks of the boundary: the shared library builds/loads without a GPU and exports exactly the entry points
that include/empose_hip.h declares; the ctypes table mirrors the header; CPU tensors are refused (no fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from em_pose_amd import _lib
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, 'include', 'empose_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(empose_[a-z0-9_]+)\s*\(', text)))


def test_header_symbols_are_exported_and_bound():
    names = _declared()
    assert 'empose_lgd_forward' in names and 'empose_smpl_sensors_fwd_bwd' in names and len(names) >= 18
    lib = _lib.lib()
    for n in names:
        assert hasattr(lib, n), 'library does not export ' + n
        assert n in _lib.SIGNATURES, 'ctypes table misses ' + n
    assert sorted(_lib.SIGNATURES) == names
    assert lib.empose_arch() == b'gfx950'


def test_struct_layouts_match_the_header_field_order():
    text = open(os.path.join(ROOT, 'include', 'empose_hip.h')).read()

    def fields(struct_name):
        end = re.search(r'\}\s*' + struct_name + ';', text).start()
        start = text.rfind('typedef struct {', 0, end) + len('typedef struct {')
        body = text[start:end]
        body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
        out = []
        for decl in body.split(';'):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(','):
                name = re.findall(r'([A-Za-z_][A-Za-z0-9_]*)\s*(?:\[[^\]]*\])*\s*$', part.strip())
                out.append(name[0])
        return out
    for cname, cls in (('empose_smpl_desc', _lib.SmplDesc), ('empose_dense_desc', _lib.DenseDesc),
                       ('empose_mlp_desc', _lib.MlpDesc), ('empose_lstm_desc', _lib.LstmDesc),
                       ('empose_model_desc', _lib.ModelDesc), ('empose_lgd_io', _lib.LgdIO),
                       ('empose_mesh_desc', _lib.MeshDesc), ('empose_rnn_desc', _lib.RnnDesc)):
        assert fields(cname) == [f[0] for f in cls._fields_], cname


def test_bad_descriptors_are_rejected_without_a_gpu():
    lib = _lib.lib()
    desc = _lib.ModelDesc()
    handle = ctypes.c_void_p()
    assert lib.empose_model_create(ctypes.byref(desc), ctypes.byref(handle)) == -1
    assert b'n_sensors' in lib.empose_last_error()
    assert lib.empose_lgd_workspace_bytes(None, 4, 4) == 0


def test_cpu_tensors_raise_instead_of_falling_back():
    from em_pose_amd.bodymodels.smpl import SMPLLayer
    from em_pose_amd.helpers.configuration import lgd_config
    from em_pose_amd.nn.models import create_model
    smpl = SMPLLayer(H.small_model())
    net = create_model(lgd_config(12, True, 2, hidden=32, rnn_hidden=32), smpl).eval()
    x = torch.zeros(1, 4, 36)
    with pytest.raises(_lib.EmposeError):
        net.forward_tensors(x, torch.zeros(1, 4, 108), torch.zeros(1, 12, 3), torch.eye(3).expand(1, 12, 3, 3))
    with pytest.raises(_lib.EmposeError):
        smpl(poses_body=torch.zeros(2, 63), betas=torch.zeros(2, 10))
    net.train()
    with pytest.raises(RuntimeError):  # the inference entry point refuses training mode ...
        net.forward_tensors(x, x, x, x)
    from em_pose_amd.data.data import SyntheticBatch
    w = {'poses': np.zeros((1, 4, 66), np.float32), 'shapes': np.zeros((1, 10), np.float32),
         'marker_pos': np.zeros((1, 4, 36), np.float32), 'marker_oris': np.zeros((1, 4, 108), np.float32),
         'offset_t': np.zeros((1, 12, 3), np.float32), 'offset_r': np.tile(np.eye(3, dtype=np.float32), (1, 12, 1, 1))}
    with pytest.raises(_lib.EmposeError):  # ... and the training path refuses CPU tensors as well
        net(SyntheticBatch(w))
