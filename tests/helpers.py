"""Shared test utilities (fixture loading, oracle plumbing)."""
import os

import numpy as np
import torch

from oracle import torch_ref as R

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    groups = {}
    for k in z.files:
        g, rest = k.split('/', 1)
        groups.setdefault(g, {})[rest] = z[k]
    return groups


def small_model():
    z = np.load(os.path.join(GOLDEN, 'smpl_small.npz'))
    return {k: z[k] for k in z.files}


def sd_to_torch(sd, dtype=torch.float32):
    out = {}
    for k, v in sd.items():
        t = torch.from_numpy(np.asarray(v))
        out[k] = t.to(dtype) if t.is_floating_point() else t
    return out


def oracle_inputs(w, sl=None, sf=None, ef=None, dtype=torch.float32):
    B, F = w['poses'].shape[:2]
    s = slice(sf, ef)
    inp = {'marker_pos': torch.from_numpy(w['marker_pos'][:, s]).to(dtype),
           'marker_oris': torch.from_numpy(w['marker_oris'][:, s]).to(dtype),
           'offset_t': torch.from_numpy(w['offset_t']).to(dtype),
           'offset_r': torch.from_numpy(w['offset_r']).to(dtype),
           'marker_masks': None}
    n = inp['marker_pos'].shape[1]
    inp['seq_lengths'] = torch.full((B,), n, dtype=torch.int64) if sl is None else torch.as_tensor(sl).long()
    if 'marker_masks' in w:
        m = torch.from_numpy(w['marker_masks'][:, s]).to(dtype)
        inp['marker_masks'] = m
        # RealBatch.get_inputs suppresses missing sensors to 0 (reference data.py:284-306)
        valid = (m == 1.0)[..., None]
        inp['marker_pos'] = (inp['marker_pos'].reshape(B, n, 12, 3) * valid).reshape(B, n, -1)
        inp['marker_oris'] = (inp['marker_oris'].reshape(B, n, 12, 9) * valid).reshape(B, n, -1)
    return inp


def run_oracle(case, tag, inp, dtype=torch.float32, state=None):
    meta = case['meta']
    model = small_model()
    bm = R.BodyModelTensors(model, dtype=dtype)
    vids = [int(v) for v in meta['vertex_ids']]
    tables = R.sensor_tables(model['f'], vids)
    sd = sd_to_torch(case['sd'], dtype)
    return R.ief_forward(sd, bm, tables, vids, inp, n_markers=int(meta['n_markers']), N=int(meta['N']),
                         rnn_init=bool(meta['rnn']), rnn_state=state)


def run_oracle_baseline(case, inp, state=None, dtype=torch.float32):
    """The ResNet / (Bi)RNN baselines through the oracle, configured from the fixture's recorded reference flags."""
    import json
    fl = json.loads(str(case['meta']['flags']))
    bm = R.BodyModelTensors(small_model(), dtype=dtype)
    sd = sd_to_torch(case['sd'], dtype)
    common = dict(n_markers=fl['n_markers'], num_layers=fl['m_num_layers'], estimate_shape=fl['m_estimate_shape'],
                  shape_avg=fl['m_average_shape'], do_fk=fl['m_fk_loss'] > 0, skip=fl.get('m_skip_connections', False))
    if fl['m_type'] == 'resnet':
        return R.resnet_forward(sd, bm, inp, **common), None
    return R.simple_rnn_forward(sd, bm, inp, bidirectional=fl.get('m_bidirectional', False),
                                learn_init_state=fl.get('m_learn_init_state', False), state=state, **common)
