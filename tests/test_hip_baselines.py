"""
GPU parity tests of the two baselines next to the LGD path (reference models.py:166-366,
layers.py:80-182): the stand-alone (Bi)LSTM entry point `empose_rnn_fwd`, the fused residual block of
`empose_linear_f32_ex`, and the `SimpleRNN` / `FeedForwardResNet` modules against the oracle and the vectors recorded
from the reference.  Tolerance as everywhere: 1e-4 abs fp32 (BASELINE.json north_star).
"""
import json

import numpy as np
import pytest
import torch

from em_pose_amd import _lib, synthetic
from em_pose_amd.bodymodels.smpl import SMPLLayer
from em_pose_amd.data.data import RealBatch
from em_pose_amd.helpers.configuration import CONSTANTS as CONST
from em_pose_amd.helpers.configuration import Configuration
from em_pose_amd.nn.layers import FeedForwardResidualBlock, RNNLayer
from em_pose_amd.nn.models import create_model
from oracle import torch_ref as R
from tests import helpers as H

pytestmark = pytest.mark.gpu
ATOL = 1e-4
DEV = 'cuda:0'


def _lstm_sd(layer):
    return {'lstm.' + k: v.detach().cpu() for k, v in layer.lstm.state_dict().items()}


@pytest.mark.parametrize('bi,L,In,Hd,B,F', [
    (False, 1, 8, 4, 1, 1),          # smallest legal shapes
    (False, 3, 72, 36, 5, 7),        # hidden size not a multiple of the 32-unit tile; three layers, small batch
    (True, 1, 12, 32, 3, 5),
    (True, 2, 144, 64, 7, 24),
    (True, 2, 144, 512, 33, 32),     # the released BiRNN width, rows not a multiple of the 64-row tile
    (True, 4, 16, 40, 4, 9),         # 8 units = the most the handle packs
    (False, 2, 144, 512, 70, 40),
    (False, 2, 72, 36, 40, 9),       # K-split kernel (17..256 rows): hidden size not a multiple of its 16-unit tile
    (True, 1, 20, 24, 257, 5),       # ... and the first batch past it (chain kernel), both directions
    (True, 2, 144, 64, 16, 9),       # the largest batch of the weight-streaming small-batch kernel ...
    (True, 2, 144, 64, 17, 9),       # ... and the smallest of the matrix-core kernel
    (False, 2, 60, 512, 1, 40),      # whole-sequence kernel (uni-directional, B <= 16): the LGD init RNN, streaming
    (False, 2, 144, 512, 16, 33),    # ... its largest batch
    (False, 4, 300, 256, 3, 12),     # ... four layers, input wider than 256
    (False, 1, 16, 8, 2, 4),         # ... one layer, shortest sequence it takes
    (False, 3, 60, 512, 2, 20),      # ... three layers (one idle wave per block)
])
def test_rnn_layer_vs_oracle_and_nn_lstm(bi, L, In, Hd, B, F):
    """empose_rnn_fwd: ragged rows, both directions, given initial state, final state (reference layers.py:133-157)."""
    torch.manual_seed(B * 1000 + F)
    layer = RNNLayer(In, Hd, L, bidirectional=bi).eval()
    with torch.no_grad():
        for p in layer.lstm.parameters():
            p.mul_(2.0)   # default init is +-1/sqrt(H): make the gates leave their linear range
    x = torch.randn(B, F, In)
    lens = torch.randint(1, F + 1, (B,))
    lens[0] = F
    U = L * (2 if bi else 1)
    h0, c0 = 0.5 * torch.randn(U, B, Hd), 0.5 * torch.randn(U, B, Hd)
    for state in (None, (h0, c0)):
        want, (wh, wc) = R.lstm_forward(_lstm_sd(layer), 'lstm.', x, lens, state, L, bi)
        # the plain PyTorch op on the same inputs (packed sequences), to pin the oracle's loop here as well
        from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence
        with torch.no_grad():
            packed = pack_padded_sequence(x, lens, batch_first=True, enforce_sorted=False)
            ref, (rh, rc) = layer.lstm(packed, state)
            ref, _ = pad_packed_sequence(ref, batch_first=True, total_length=F)
        np.testing.assert_allclose(want.numpy(), ref.numpy(), atol=2e-5)
        np.testing.assert_allclose(wh.numpy(), rh.numpy(), atol=2e-5)

        g = layer.to(DEV)
        g.init_state = None if state is None else tuple(t.to(DEV) for t in state)
        got = g(x.to(DEV), lens.to(DEV))
        torch.cuda.synchronize()
        np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), atol=ATOL)
        np.testing.assert_allclose(g.final_state[0].cpu().numpy(), wh.numpy(), atol=ATOL)
        np.testing.assert_allclose(g.final_state[1].cpu().numpy(), wc.numpy(), atol=ATOL)
        layer = g.cpu()
    layer.release()


def test_whole_sequence_kernel_equals_step_launches(monkeypatch):
    """B <= 16 uni-directional stacks run the whole sequence in one cooperative launch (lstm_persist_kernel); the same
    call stepped launch by launch (lstm_small_kernel, empose_set_option("lstm_persist", 0)) gives the same bits: outputs,
    final state, ragged rows, state carry over two chunks."""
    torch.manual_seed(5)
    # (round 5: from 4 rows on both modes would run lstm_fewrows_kernel, whose sums are ordered differently; this test is
    # about the two kernels that share their bits -- tests/test_hip_round5.py holds the new one against them)
    _lib.check(_lib.lib().empose_set_option(b'lstm_fewrows', 0))
    layer = RNNLayer(60, 512, 2).eval()
    x = torch.randn(6, 64, 60)
    lens = torch.tensor([64, 1, 33, 64, 17, 2])
    res = {}
    for mode in ('1', '0'):
        _lib.check(_lib.lib().empose_set_option(b'lstm_persist', int(mode)))
        g = layer.to(DEV)
        g.init_state = None
        outs = []
        for chunk in range(2):
            y = g(x[:, chunk * 32:(chunk + 1) * 32].contiguous().to(DEV), (lens - chunk * 32).clamp(0, 32).clamp(min=1).to(DEV))
            g.init_state = g.final_state
            outs += [y.cpu(), g.final_state[0].cpu(), g.final_state[1].cpu()]
        res[mode] = outs
        layer = g.cpu()
    for p_, s_ in zip(res['1'], res['0']):
        assert torch.isfinite(p_).all()
        assert torch.equal(p_, s_)
    layer.release()


def test_rnn_layer_rejects_what_it_cannot_do():
    layer = RNNLayer(16, 8, 1, bidirectional=True).eval()
    with pytest.raises(_lib.EmposeError):
        layer(torch.zeros(1, 2, 16), torch.tensor([2]))        # CPU tensors: no fallback
    assert isinstance(RNNLayer(16, 8, 1, dropout=0.1).input_drop, torch.nn.Dropout)   # (round 6: a training-time feature)
    with pytest.raises(NotImplementedError):
        RNNLayer(16, 8, 1, bidirectional=True, learn_init_state=True)
    lib = _lib.lib()
    desc = _lib.RnnDesc()
    desc.num_layers, desc.input_size, desc.hidden_size, desc.bidirectional = 5, 16, 8, 1
    handle = _lib.C.c_void_p()
    assert lib.empose_rnn_create(_lib.C.byref(desc), _lib.C.byref(handle)) != 0  # 10 units > 8
    desc.num_layers, desc.hidden_size = 1, 6
    assert lib.empose_rnn_create(_lib.C.byref(desc), _lib.C.byref(handle)) != 0  # hidden % 4


@pytest.mark.parametrize('M,Hd', [(1, 4), (77, 64), (1000, 512)])
def test_residual_block_vs_torch(M, Hd):
    """relu(W x + b + x) in one launch (reference layers.py:170-182)."""
    torch.manual_seed(M)
    blk = FeedForwardResidualBlock(Hd, Hd).eval()
    x = torch.randn(M, Hd)
    with torch.no_grad():
        want = torch.relu(x @ blk.dense.weight.t() + blk.dense.bias + x)
    got = blk.to(DEV)(x.to(DEV))
    torch.cuda.synchronize()
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), atol=ATOL)
    assert (got >= 0).all()


def _build(case):
    fl = json.loads(str(case['meta']['flags']))
    net = create_model(Configuration.defaults(**fl), SMPLLayer(H.small_model()))
    missing, unexpected = net.load_state_dict(H.sd_to_torch(case['sd']), strict=False)
    assert not unexpected and all(k.startswith('smpl.') for k in missing)
    assert net.model_name() == str(case['meta']['model_name'])
    return net.to(DEV).eval(), fl


def _batch(w, rec, lengths, masks, sf=None, ef=None):
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    B = w['poses'].shape[0]
    F = w['poses'][:, sf:ef].shape[1]
    masks = np.ones((B, F, 12), dtype=np.float32) if masks is None else masks[:, sf:ef]
    b = RealBatch(list(range(B)), torch.as_tensor(lengths), t(w['poses'][:, sf:ef]), t(w['shapes']),
                  torch.zeros(B, F, 3), t(w['marker_pos'][:, sf:ef]), t(w['marker_oris'][:, sf:ef]), t(masks),
                  t(w['offset_t']), t(w['offset_r'])).to_gpu(torch.device(DEV))
    b.joints_gt = t(rec['joints_gt']).to(DEV)
    return b


def _check(net, rec, out, loss_vals):
    for k in ('pose_hat', 'root_ori_hat', 'shape_hat', 'joints_hat'):
        assert out[k].shape == rec['out_' + k].shape
        np.testing.assert_allclose(out[k].cpu().numpy(), rec['out_' + k], atol=ATOL, err_msg=k)
    for k in ('pose', 'root_pose', 'shape', 'fk', 'total_loss'):
        np.testing.assert_allclose(loss_vals[k], float(rec['loss_' + k]), rtol=1e-4, atol=1e-5, err_msg=k)
    if 'rnn_h' in rec:
        np.testing.assert_allclose(net.rnn.final_state[0].cpu().numpy(), rec['rnn_h'], atol=ATOL)
        np.testing.assert_allclose(net.rnn.final_state[1].cpu().numpy(), rec['rnn_c'], atol=ATOL)


@pytest.mark.parametrize('name', ['birnn12_ragged', 'resnet12'])
def test_golden_baselines_ragged_masked(name):
    """forward(batch) + loss values of the reference's baselines on a ragged batch with missing sensors."""
    case = H.load_case(name)
    net, _ = _build(case)
    w = case['in']
    b = _batch(w, case['run'], w['seq_lengths'], w['marker_masks'])
    out = net(b)
    _, loss_vals = net.backward(b, out)
    _check(net, case['run'], out, loss_vals)


@pytest.mark.parametrize('name', ['rnn12_carry', 'rnn6_learninit_carry'])
def test_golden_rnn_baselines_state_carry(name):
    """Two consecutive chunks: `is_new_sequence=False` carries (h, c); a learned initial state replaces it."""
    case = H.load_case(name)
    net, _ = _build(case)
    for tag, (sf, ef), new in (('chunk0', (0, 24), True), ('chunk1', (24, 48), False)):
        b = _batch(case['in'], case[tag], [24, 24], None, sf, ef)
        out = net(b, is_new_sequence=new)
        _, loss_vals = net.backward(b, out)
        _check(net, case[tag], out, loss_vals)


@pytest.mark.parametrize('name', ['train_birnn12', 'train_rnn6_l3', 'train_resnet12', 'train_resnet6_nofk_noshape',
                                  'train_rnn6_learninit'])
def test_golden_baselines_training_step(name):
    """One training step of the baselines -- train mode, `forward(batch)`, `backward(batch, out)` = the losses and
    `total_loss.backward()` (reference models.py:196-262, 297-366) -- on a ragged batch with missing sensors, against what
    the unmodified reference deposited: outputs, loss values and EVERY parameter gradient.  The graph runs over the HIP
    kernels: linear layers and their reverse (matrix-core GEMM / A^T B), the LSTM with back-propagation through time (a
    bidirectional stack composed per layer and direction; a learned initial state through the cotangents of (h_0, c_0)),
    joints with the sub-mesh vector-Jacobian product."""
    case = H.load_case(name)
    net, fl = _build(case)
    net.fk_vertex_ids = [int(v) for v in case['meta']['vertex_ids']]
    net.train()
    w, rec = case['in'], case['run']
    b = _batch(w, rec, w['seq_lengths'], w['marker_masks'])
    net.zero_grad()
    out = net(b)
    total, loss_vals = net.backward(b, out)
    torch.cuda.synchronize()
    for k in ('pose_hat', 'root_ori_hat', 'shape_hat', 'joints_hat'):
        if ('out_' + k) in rec:
            np.testing.assert_allclose(out[k].detach().cpu().numpy(), rec['out_' + k], atol=ATOL, err_msg=k)
        else:
            assert out[k] is None
    for k in ('pose', 'root_pose', 'shape', 'fk', 'total_loss'):
        np.testing.assert_allclose(loss_vals[k], float(rec['loss_' + k]), rtol=1e-4, atol=1e-5, err_msg=k)
    params = dict(net.named_parameters())
    names = [k[len('grad/'):] for k in rec if k.startswith('grad/')]
    assert len(names) == len([n for n, p in params.items() if not n.startswith('smpl.') and p.requires_grad])
    worst = 0.0
    for n in names:
        want, got = rec['grad/' + n], params[n].grad
        assert got is not None, n
        scale = max(float(np.abs(want).max()), 1e-3)
        err = float(np.abs(got.cpu().numpy() - want).max()) / scale
        worst = max(worst, err)
        assert err <= 1e-4, (n, err)
    print('%s: %d gradients, worst error / max-abs of the tensor %.2e' % (name, len(names), worst))
    # eval mode afterwards: the inference kernels, no graph
    net.eval()
    out = net(b)
    assert not out['pose_hat'].requires_grad


def test_training_mode_dropout_is_applied_and_eval_ignores_it():
    """`m_dropout` (inputs of the RNN) and `m_dropout_hidden` (the shape MLP): every released configuration uses 0.0, the
    reference's flags exist (configuration.py:164,172; layers.py:30,62,103).  Training mode: two forwards differ (masks are
    drawn) and the gradients are finite; eval mode: deterministic and equal to the same network without dropout."""
    flags = dict(m_type='rnn', m_hidden_size=32, m_num_layers=2, m_bidirectional=True, window_size=32, use_marker_pos=True,
                 use_marker_ori=True, m_estimate_shape=True, m_shape_hidden_size=24, m_average_shape=False, m_fk_loss=0.0,
                 n_markers=12, m_dropout=0.3, m_dropout_hidden=0.4)
    torch.manual_seed(9)
    net = create_model(Configuration.defaults(**flags), SMPLLayer(H.small_model())).to(DEV)
    plain = create_model(Configuration.defaults(**dict(flags, m_dropout=0.0, m_dropout_hidden=0.0)),
                         SMPLLayer(H.small_model())).to(DEV)
    plain.load_state_dict(net.state_dict())
    case = H.load_case('train_birnn12')
    w, rec = case['in'], case['run']
    b = _batch(w, rec, w['seq_lengths'], w['marker_masks'])
    net.eval(); plain.eval()
    a1, a2, a3 = net(b), net(b), plain(b)
    assert torch.equal(a1['pose_hat'], a2['pose_hat']) and torch.equal(a1['pose_hat'], a3['pose_hat'])
    assert torch.equal(a1['shape_hat'], a3['shape_hat'])
    net.train()
    t1, t2 = net(b), net(b)
    assert not torch.equal(t1['pose_hat'], t2['pose_hat']) and not torch.equal(t1['shape_hat'], t2['shape_hat'])
    net.zero_grad()
    total, _ = net.backward(b, t2)
    for n, p in net.named_parameters():
        if not n.startswith('smpl.'):
            assert p.grad is not None and bool(torch.isfinite(p.grad).all()), n


@pytest.mark.parametrize('m_type', ['rnn', 'resnet'])
def test_full_size_baselines_vs_oracle(m_type):
    """The released widths (BiRNN 2x512, ResNet 512) on the V=6890 body model against the oracle."""
    model = synthetic.make_model()
    flags = dict(m_type=m_type, m_hidden_size=512, m_num_layers=2, m_bidirectional=(m_type == 'rnn'), window_size=32,
                 use_marker_pos=True, use_marker_ori=True, m_estimate_shape=True, m_shape_hidden_size=128,
                 m_average_shape=True, m_fk_loss=0.1, n_markers=12)
    torch.manual_seed(3)
    net = create_model(Configuration.defaults(**flags), SMPLLayer(model)).eval()
    B, F = 6, 32
    bm = R.BodyModelTensors(model)
    tables = R.sensor_tables(model['f'], CONST.VERTEX_IDS)

    def sensors(poses, betas, o_r, o_t):
        with torch.no_grad():
            p, o, _ = R.estimated_markers(bm, tables, CONST.VERTEX_IDS, torch.from_numpy(poses),
                                          torch.from_numpy(betas), torch.from_numpy(o_r), torch.from_numpy(o_t))
        return p.numpy(), o.numpy()
    w = synthetic.make_windows(B, F, 11, sensors)
    lens = torch.tensor([32, 32, 20, 32, 1, 9])
    inp = H.oracle_inputs(w, sl=lens)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items() if not k.startswith('smpl.')}
    kw = dict(n_markers=12, num_layers=2, estimate_shape=True, shape_avg=True, do_fk=True)
    if m_type == 'rnn':
        want, _ = R.simple_rnn_forward(sd, bm, inp, bidirectional=True, **kw)
    else:
        want = R.resnet_forward(sd, bm, inp, **kw)
    net = net.to(DEV)
    b = _batch(w, {'joints_gt': np.zeros((B, F, 66), np.float32)}, lens, None)
    out = net(b)
    torch.cuda.synchronize()
    for k in ('pose_hat', 'root_ori_hat', 'shape_hat', 'joints_hat'):
        np.testing.assert_allclose(out[k].cpu().numpy(), want[k].numpy(), atol=ATOL, err_msg=k)
