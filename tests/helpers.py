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


# ----------------------------------------------------------------------------------------------------------------------
# Full-width training fixtures (tests/golden/train_fp_*.npz): 6 M parameters cannot be stored, so the weights are a
# deterministic function of (name, shape, seed) on torch's CPU generator -- the generating script (reference network) and
# the GPU test (this repository's network) both call `seeded_state_dict` -- and the fixture holds FINGERPRINTS of every
# parameter gradient instead of the gradients.
# ----------------------------------------------------------------------------------------------------------------------
def seeded_state_dict(template, seed):
    """:param template: a state_dict (names + shapes; values ignored).  Body-model entries are skipped."""
    import zlib
    out = {}
    for k, v in template.items():
        if k.startswith('smpl.') or k.endswith('num_batches_tracked'):
            continue
        # seeded by NAME: positions differ between the reference's module and this repository's (body-model entries)
        g = torch.Generator().manual_seed(int(seed) * 1000003 + zlib.crc32(k.encode()) % 1000003)
        shape = tuple(v.shape)
        if 'running_var' in k:
            t = torch.rand(shape, generator=g) + 0.5
        elif 'running_mean' in k:
            t = torch.randn(shape, generator=g) * 0.1
        elif 'batch_norm' in k or (k.split('.')[-2].isdigit() and len(shape) == 1 and '.layers.' in k
                                   and int(k.split('.')[-2]) in (1, 5)):
            t = (torch.rand(shape, generator=g) + 0.5) if k.endswith('weight') else torch.randn(shape, generator=g) * 0.1
        elif shape == (1,):                       # PReLU slope
            t = torch.full(shape, 0.25) + 0.1 * torch.rand(shape, generator=g)
        else:
            fan_in = shape[-1] if len(shape) > 1 else 512
            t = (torch.rand(shape, generator=g) * 2.0 - 1.0) / float(np.sqrt(fan_in))
            if 'hidden_to_output' in k or k.startswith(('pose_net_init', 'shape_net_init')):
                t = t * 3.0
        out[k] = t.to(torch.float32)
    return out


N_PROJ, N_SAMPLE = 8, 256


def tensor_fingerprint(name, t):
    """max-abs, L2 norm, N_PROJ seeded Gaussian projections and N_SAMPLE seeded entries of a tensor (float64); the seed is
    a function of the tensor's NAME (the same in the reference's module and in this repository's)."""
    import zlib
    x = torch.as_tensor(t).detach().to('cpu', torch.float64).reshape(-1)
    z, idx = fingerprint_basis(name, x.numel())
    return {'max': float(x.abs().max()), 'l2': float(x.norm()), 'n': int(x.numel()), 'proj': (z @ x).numpy(),
            'sample': x[idx].numpy()}


def fingerprint_basis(name, numel):
    """The N_PROJ Gaussian directions (N_PROJ, numel) and the N_SAMPLE entry indices `tensor_fingerprint` uses for a tensor
    of that name and size."""
    import zlib
    g = torch.Generator().manual_seed(770000 + zlib.crc32(name.encode()) % 100000)
    z = torch.randn(N_PROJ, numel, generator=g, dtype=torch.float64)
    idx = torch.randperm(numel, generator=g)[:N_SAMPLE]
    return z, idx


def batch_norm_of_prelu(name):
    """Module name of the BatchNorm1d in front of a PReLU of an MLP (reference nn/layers.py:13-77): `X.activation_fn` ->
    `X.batch_norm`; `X.hidden_layers.j.layers.2` -> `.layers.1`; `.layers.6` -> `.layers.5`."""
    if name.endswith('.activation_fn'):
        return name[:-len('activation_fn')] + 'batch_norm'
    head, i = name.rsplit('.', 1)
    assert int(i) in (2, 6), name
    return '{}.{}'.format(head, int(i) - 1)


def prelu_of_batch_norm(name):
    if name.endswith('.batch_norm'):
        return name[:-len('batch_norm')] + 'activation_fn'
    head, i = name.rsplit('.', 1)
    assert int(i) in (1, 5), name
    return '{}.{}'.format(head, int(i) + 1)


def explain_by_prelu_flips(fp, name, mine_full, want, tol_e, tol_p, tol_l, slope):
    """PROOF that what separates a BatchNorm parameter gradient from the reference's fingerprint is a PReLU branch flip.

    `fp` is a tests/golden/train_fp_*.npz fixture: for every PReLU application of the reference's step it holds the
    elements nearest to zero -- (row, column, z, summed cotangent of the PReLU output, normalised BatchNorm input x^) --
    and `z_noise`, how far the reference's own five fp32 realisations move that application's z.  An element with
    |z| <= z_noise may legitimately sit on the other side of zero in another fp32 implementation; if it does, the gradient
    of the BatchNorm bias in front of that PReLU moves at THAT column by exactly  -sign(z) (1 - slope) cot  (its weight:
    times x^), nowhere else.  Accepted only if one or two such recorded elements reproduce the observed difference: every
    sampled entry, all N_PROJ projections and the norm must be within their ORDINARY tolerances once the predicted
    difference is subtracted.  Returns the list of (call, row, column, z, predicted difference) or None."""
    import itertools
    kind = name.rsplit('.', 1)[1]
    if kind not in ('weight', 'bias'):
        return None
    try:
        prelu = prelu_of_batch_norm(name.rsplit('.', 1)[0])
    except (AssertionError, ValueError):
        return None
    cands = []
    c = 0
    while 'prelu/{}/{}/z'.format(prelu, c) in fp:
        key = 'prelu/{}/{}/'.format(prelu, c)
        noise = float(fp[key + 'z_noise'])
        for row, col, z, cot, xhat in zip(fp[key + 'row'], fp[key + 'col'], fp[key + 'z'], fp[key + 'cot'], fp[key + 'xhat']):
            if abs(z) <= noise and cot != 0.0:
                pred = -np.sign(z) * (1.0 - slope) * cot * (xhat if kind == 'weight' else 1.0)
                cands.append((c, int(row), int(col), float(z), float(pred)))
        c += 1
    n = int(want['n'])
    Z, idx = fingerprint_basis(name, n)
    Z, idx = Z.numpy(), idx.numpy()
    mine_full = np.asarray(mine_full, dtype=np.float64).reshape(-1)
    ds, dp = mine_full[idx] - want['sample'], Z @ mine_full - want['proj']
    for size in (1, 2):
        for sub in itertools.combinations(cands, size):
            d = np.zeros(n)
            for _, _, col, _, pred in sub:
                d[col] += pred
            if np.abs(ds - d[idx]).max() > tol_e or np.abs(dp - Z @ d).max() > tol_p:
                continue
            # the reference's norm with the flips applied: ||want + d||^2 = ||want||^2 + sum_c (2 want_c d_c + d_c^2),
            # want_c = mine_c - d_c up to the tolerance just checked
            cols = np.nonzero(d)[0]
            l2 = np.sqrt(max(float(want['l2']) ** 2 + float(np.sum(2.0 * (mine_full[cols] - d[cols]) * d[cols] + d[cols] ** 2)), 0.0))
            if abs(np.linalg.norm(mine_full) - l2) <= tol_l + 2.0 * tol_e * size:
                return list(sub)
    return None
