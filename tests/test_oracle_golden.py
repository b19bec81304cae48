"""
Pins the oracle (oracle/torch_ref.py) against vectors recorded from the unmodified reference
(tests/golden/make_golden.py).  CPU only.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import torch_ref as R
from tests import helpers as H

TOL = 2e-5  # fp32 restatement vs fp32 reference, different op order


def _check(rec, out, hist, tol=TOL):
    for k in ('pose_hat', 'root_ori_hat', 'shape_hat', 'joints_hat'):
        np.testing.assert_allclose(out[k].numpy(), rec['out_' + k], atol=tol, rtol=0)
    B, F = rec['out_pose_hat'].shape[:2]
    for name in ('pose', 'shape', 'joints', 'markers', 'markers_ori'):
        mine = np.stack([t.numpy().reshape(B, F, -1) for t in hist[name]])
        np.testing.assert_allclose(mine, rec['hist_' + name], atol=tol, rtol=0)
    # gradient features are O(10): relative tolerance
    gp = np.stack([t.numpy() for t in hist['g_pose']])
    gs = np.stack([t.numpy() for t in hist['g_shape']])
    np.testing.assert_allclose(gp, rec['g_pose'], atol=5e-4, rtol=1e-4)
    np.testing.assert_allclose(gs, rec['g_shape'], atol=5e-4, rtol=1e-4)


def test_known_answers():
    with open(os.path.join(H.GOLDEN, 'known_answers.json')) as f:
        known = json.load(f)
    # reference README.md:228: 5 721 419 trainable parameters = 5 721 250 network + 169 BodyModel parameters
    assert known['lgd_rnn_6_N2']['params_without_bodymodel'] + 169 == 5721419
    assert known['lgd_rnn_6_N2']['model_name'] == 'IEF-2x512-N2-RNN-2x512-r0.01-ws32-lr0.0005-grad-n6'
    assert known['lgd_12_N4']['model_name'] == 'IEF-2x512-N4-r0.01-ws32-lr0.001-grad-n12'


def test_lgd12_no_rnn():
    case = H.load_case('lgd12_n4')
    out, hist = H.run_oracle(case, 'run', H.oracle_inputs(case['in']))
    _check(case['run'], out, hist)


def test_lgdrnn12_carry():
    case = H.load_case('lgdrnn12_n4_carry')
    out0, hist0 = H.run_oracle(case, 'chunk0', H.oracle_inputs(case['in'], sf=0, ef=32))
    _check(case['chunk0'], out0, hist0)
    np.testing.assert_allclose(hist0['rnn_state'][0].numpy(), case['chunk0']['rnn_h'], atol=TOL)
    np.testing.assert_allclose(hist0['rnn_state'][1].numpy(), case['chunk0']['rnn_c'], atol=TOL)
    out1, hist1 = H.run_oracle(case, 'chunk1', H.oracle_inputs(case['in'], sf=32, ef=64), state=hist0['rnn_state'])
    _check(case['chunk1'], out1, hist1)


def test_lgdrnn6():
    case = H.load_case('lgdrnn6_n2')
    out, hist = H.run_oracle(case, 'run', H.oracle_inputs(case['in']))
    _check(case['run'], out, hist)


def test_ragged_masked():
    case = H.load_case('lgdrnn12_n3_ragged_masked')
    out, hist = H.run_oracle(case, 'run', H.oracle_inputs(case['in'], sl=case['in']['seq_lengths']))
    _check(case['run'], out, hist)
    np.testing.assert_allclose(hist['rnn_state'][0].numpy(), case['run']['rnn_h'], atol=TOL)


def test_inner_windows():
    """`forward(batch, window_size=k)` on one sequence (reference models.py:146-163): inner windows of 16, 16, 8 frames,
    then a second call with 12, 12, 8 frames continuing the LSTM state.  The oracle runs one `ief_forward` per inner
    window with the state handed on; concatenated along time it equals what the reference's single call returned."""
    case = H.load_case('lgdrnn12_n4_inner_windows')
    state = None
    for tag, (sf, ef) in (('call0', (0, 40)), ('call1', (40, 72))):
        rec = case[tag]
        ws = int(rec['window_size'])
        outs, hists = [], []
        for a in range(sf, ef, ws):
            out, hist = H.run_oracle(case, tag, H.oracle_inputs(case['in'], sf=a, ef=min(a + ws, ef)), state=state)
            state = hist['rnn_state']
            outs.append(out)
            hists.append(hist)
        out = {k: torch.cat([o[k] for o in outs], dim=1) for k in outs[0]}
        hist = {}
        for name in ('pose', 'shape', 'joints', 'markers', 'markers_ori'):
            hist[name] = [torch.cat([h[name][n].reshape(1, min(ws, ef - sf - i * ws), -1) for i, h in enumerate(hists)],
                                    dim=1) for n in range(len(hists[0][name]))]
        for name in ('g_pose', 'g_shape'):
            hist[name] = [torch.cat([h[name][n] for h in hists], dim=0) for n in range(len(hists[0][name]))]
        _check(rec, out, hist)
        np.testing.assert_allclose(state[0].numpy(), rec['rnn_h'], atol=TOL)
        np.testing.assert_allclose(state[1].numpy(), rec['rnn_c'], atol=TOL)


def test_components():
    z = np.load(os.path.join(H.GOLDEN, 'components.npz'))
    gt, hat = torch.from_numpy(z['rl_gt']), torch.from_numpy(z['rl_hat'])
    sl, mm = torch.from_numpy(z['rl_len']), torch.from_numpy(z['rl_mask'])
    np.testing.assert_allclose(R.reconstruction_loss(gt, hat).numpy(), z['rl_plain'], rtol=1e-6)
    np.testing.assert_allclose(R.reconstruction_loss(gt, hat, sl).numpy(), z['rl_len_only'], rtol=1e-6)
    np.testing.assert_allclose(R.reconstruction_loss(gt, hat, sl, mm).numpy(), z['rl_full'], rtol=1e-6)
    assert (R.mask_from_seq_lengths(sl).numpy() == z['mask_from_len']).all()

    model = H.small_model()
    vids = [int(v) for v in H.load_case('lgd12_n4')['meta']['vertex_ids']]
    tables = R.sensor_tables(model['f'], vids)
    assert (tables[0] == z['vs_sub_faces']).all()
    assert (tables[1] == z['vs_sub_vertex_faces']).all()
    assert (tables[2] == z['vs_helpers']).all()
    pos, ori, nor = R.virtual_pos_and_rot(torch.from_numpy(z['vs_verts']), vids, tables)
    np.testing.assert_allclose(pos.numpy(), z['vs_pos'], atol=1e-7)
    np.testing.assert_allclose(ori.numpy(), z['vs_ori'], atol=2e-6)
    np.testing.assert_allclose(nor.numpy(), z['vs_nor'], atol=1e-8)

    bm = R.BodyModelTensors(model)
    v, j = R.smpl_fk(bm, torch.from_numpy(z['fk_pose']), torch.from_numpy(z['fk_betas']),
                     torch.from_numpy(z['fk_root']))
    np.testing.assert_allclose(v.numpy(), z['fk_v'], atol=1e-6)
    np.testing.assert_allclose(j.numpy(), z['fk_j'], atol=1e-6)
    v, j = R.smpl_fk(bm, torch.from_numpy(z['fk_pose']), torch.from_numpy(z['fk_betas'][0]))
    np.testing.assert_allclose(v.numpy(), z['fk_v_noroot_bcast'], atol=1e-6)


def _check_baseline(rec, out, final, tol=TOL):
    for k in ('pose_hat', 'root_ori_hat', 'shape_hat', 'joints_hat'):
        np.testing.assert_allclose(out[k].numpy(), rec['out_' + k], atol=tol, rtol=0, err_msg=k)
    if final is not None:
        np.testing.assert_allclose(final[0].numpy(), rec['rnn_h'], atol=tol, rtol=0)
        np.testing.assert_allclose(final[1].numpy(), rec['rnn_c'], atol=tol, rtol=0)


def test_baselines_ragged():
    """ResNet and BiRNN baselines (reference models.py:166-366) on a ragged, masked batch."""
    for name in ('birnn12_ragged', 'resnet12'):
        case = H.load_case(name)
        inp = H.oracle_inputs(case['in'], sl=case['in']['seq_lengths'])
        out, final = H.run_oracle_baseline(case, inp)
        _check_baseline(case['run'], out, final)


def test_baselines_carry():
    """Uni-directional RNN baselines over two chunks: carried state, and the learned initial state, which replaces
    the carried one (reference layers.py:121-131, models.py:298-302)."""
    for name in ('rnn12_carry', 'rnn6_learninit_carry'):
        case = H.load_case(name)
        out0, st = H.run_oracle_baseline(case, H.oracle_inputs(case['in'], sf=0, ef=24))
        _check_baseline(case['chunk0'], out0, st)
        out1, st1 = H.run_oracle_baseline(case, H.oracle_inputs(case['in'], sf=24, ef=48), state=st)
        _check_baseline(case['chunk1'], out1, st1)


@pytest.mark.parametrize('name', ['train_birnn12', 'train_rnn6_l3', 'train_resnet12', 'train_resnet6_nofk_noshape',
                                  'train_rnn6_learninit'])
def test_baselines_training_step(name):
    """One training step of the reference's baselines (train mode, `forward`, `backward` = the losses and
    `total_loss.backward()`; reference models.py:196-262, 297-366) on a ragged batch with missing sensors: the oracle's
    restatement with autograd through its own LSTM loop, body model and losses reproduces the outputs, the loss values
    and EVERY parameter gradient the reference deposited."""
    import json
    case = H.load_case(name)
    fl = json.loads(str(case['meta']['flags']))
    w, rec = case['in'], case['run']
    inp = H.oracle_inputs(w, sl=w['seq_lengths'])
    sd = {k: v.clone().requires_grad_(True) for k, v in H.sd_to_torch(case['sd'], torch.float32).items()}
    bm = R.BodyModelTensors(H.small_model(), dtype=torch.float32)
    common = dict(n_markers=fl['n_markers'], num_layers=fl['m_num_layers'], estimate_shape=fl['m_estimate_shape'],
                  shape_avg=fl['m_average_shape'], do_fk=fl['m_fk_loss'] > 0, skip=fl.get('m_skip_connections', False))
    if fl['m_type'] == 'resnet':
        out = R.resnet_forward(sd, bm, inp, **common)
    else:
        out, _ = R.simple_rnn_forward(sd, bm, inp, bidirectional=fl.get('m_bidirectional', False),
                                      learn_init_state=fl.get('m_learn_init_state', False), **common)
    for k in ('pose_hat', 'root_ori_hat', 'shape_hat', 'joints_hat'):
        if out[k] is not None:
            np.testing.assert_allclose(out[k].detach().numpy(), rec['out_' + k], atol=TOL, rtol=0, err_msg=k)
    vals = R.baseline_losses(out, torch.from_numpy(w['poses']), torch.from_numpy(w['shapes']),
                             torch.from_numpy(rec['joints_gt']), inp['seq_lengths'], inp['marker_masks'], fl['m_fk_loss'])
    for k, v in vals.items():
        np.testing.assert_allclose(float(v.detach()), float(rec['loss_' + k]), rtol=1e-5, atol=1e-6, err_msg=k)
    vals['total_loss'].backward()
    names = [k[len('grad/'):] for k in rec if k.startswith('grad/')]
    assert names
    for n in names:
        want = rec['grad/' + n]
        got = sd[n].grad
        assert got is not None, n
        scale = max(float(np.abs(want).max()), 1e-3)
        np.testing.assert_allclose(got.numpy(), want, atol=2e-5 * scale, rtol=0, err_msg=n)


EVAL_ASSETS = os.path.join(H.GOLDEN, 'eval_assets')


@pytest.mark.parametrize('tag', ['lgdrnn6_test_real', 'lgdrnn6_hold_out', 'lgdrnn12_hold_out'])
def test_evaluate_real_entry_point(tag):
    """The reference's scripts/evaluate_real.py::main ran on tests/golden/eval_assets/ (config.json + model.pth written
    by its own writers, `*_clean.npz` recordings of 70 / 256 / 300 / 520 and 40 / 270 frames with missing sensors);
    what it printed and its model returned per 256-frame chunk is the fixture.  The oracle's restatement of that entry
    point (oracle/eval_ref.py) reproduces every row of the table and every chunk's outputs."""
    from oracle import eval_ref as E
    with open(os.path.join(EVAL_ASSETS, 'expected.json')) as f:
        exp = json.load(f)
    z = np.load(os.path.join(EVAL_ASSETS, 'expected.npz'))
    e = exp[tag]
    assert e['headers'][2:] == E.HEADERS
    rows, outs = E.evaluate_real(os.path.join(EVAL_ASSETS, 'experiments'), os.path.join(EVAL_ASSETS, 'data_real'),
                                 os.path.join(H.GOLDEN, 'smpl_small.npz'), exp['vertex_ids'], e['model_id'],
                                 e['cross_subject'])
    assert [r[0] for r in rows] == [r[1] for r in e['rows']]
    for r, w in zip(rows, e['rows']):
        np.testing.assert_allclose(r[1:], w[2:], atol=1e-3, rtol=0, err_msg=r[0])    # mm / degrees
    n_chunks = 0
    for s, seq in enumerate(outs):
        for c, o in enumerate(seq):
            for k, v in o.items():
                np.testing.assert_allclose(v, z['{}/seq{}/chunk{}/{}'.format(tag, s, c, k)], atol=TOL, rtol=0)
            n_chunks += 1
    assert n_chunks == len([k for k in z.files if k.startswith(tag + '/') and k.endswith('/pose_hat')])
