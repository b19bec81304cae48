"""
GPU tests added in round 4: the status path of the cooperative whole-sequence LSTM kernels -- a poll
that gives up poisons the outputs with NaN AND is reported (EMPOSE_ETIMEOUT from the next recurrence call /
empose_async_status) instead of returning 0 and letting NaNs be averaged into a metrics table -- and the single LDS
layout definition the row-block kernels and their launchers share.
Reference semantics being replaced: reference nn/layers.py:133-157 (RNNLayer.forward; one Python thread, no such failure).
"""
import numpy as np
import pytest
import torch

from em_pose_amd import _lib
from em_pose_amd.nn.layers import RNNLayer
from oracle import torch_ref as R

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
ETIMEOUT = -4


def _layer(In, H, L, seed):
    torch.manual_seed(seed)
    layer = RNNLayer(In, H, L).eval()
    sd = {'lstm.' + k: v.detach().clone() for k, v in layer.lstm.state_dict().items()}
    return layer.to(DEV), sd


@pytest.mark.provokes_poll_timeout
def test_poll_timeout_of_the_whole_sequence_lstm_is_reported_not_silent():
    """spin_limit = 1: the polls of the small-batch whole-sequence kernel (lstm_persist, B <= 16) give up almost at once.
    The call itself is asynchronous and returns 0; after a synchronisation empose_async_status() says EMPOSE_ETIMEOUT
    exactly when the outputs hold NaN, the NEXT recurrence call refuses with the same code (once), and with the normal
    limit the same layer gives the oracle's numbers again."""
    lib = _lib.lib()
    assert lib.empose_async_status() == 0
    B, F, In, H, L = 3, 48, 64, 128, 2     # (from 4 rows on the steps are launches of lstm_fewrows_kernel: nothing polls)
    g, sd = _layer(In, H, L, 11)
    x = torch.randn(B, F, In)
    lens = torch.full((B,), F, dtype=torch.int64)
    with torch.no_grad():
        want, _ = R.lstm_forward(sd, 'lstm.', x, lens, None, L, False)
    timed_out = 0
    for attempt in range(12):      # (whether a first re-check finds the word depends on the clocks of the moment: 4 were too few once)
        if timed_out >= 2:
            break
        _lib.check(lib.empose_set_option(b'spin_limit', 1))
        g.init_state = None
        got = g(x.to(DEV), lens.to(DEV))          # returns normally: the failure happens later, on the device
        torch.cuda.synchronize()
        _lib.check(lib.empose_set_option(b'spin_limit', 0))
        has_nan = bool(torch.isnan(got).any()) or bool(torch.isnan(g.final_state[0]).any())
        status = lib.empose_async_status()
        assert (status == ETIMEOUT) == has_nan, (status, has_nan)
        if status == ETIMEOUT:
            timed_out += 1
            assert b'timed out' in lib.empose_last_error()
            assert lib.empose_async_status() == 0       # reported once
        else:
            np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), atol=1e-4)
    assert timed_out >= 1, 'spin_limit = 1 never made a poll give up: the test does not exercise the status path'

    # the other way of learning about it: the next call that runs a recurrence refuses
    _lib.check(lib.empose_set_option(b'spin_limit', 1))
    refused = False
    for attempt in range(12):
        g.init_state = None
        try:
            g(x.to(DEV), lens.to(DEV))
        except _lib.EmposeError as e:       # the launch before this one gave up on a poll
            assert 'error -4' in str(e) and 'timed out' in str(e)
            refused = True
            break
        torch.cuda.synchronize()
    _lib.check(lib.empose_set_option(b'spin_limit', 0))
    assert refused, 'no poll gave up in 12 launches with spin_limit = 1'
    torch.cuda.synchronize()
    # STICKY: the refusal does not clear the report -- another recurrence (of any model or stream of the process)
    # is refused too, until empose_async_status() has reported it
    g.init_state = None
    with pytest.raises(_lib.EmposeError, match='timed out'):
        g(x.to(DEV), lens.to(DEV))
    assert lib.empose_async_status() == ETIMEOUT
    assert lib.empose_async_status() == 0           # reported and cleared; nothing was launched since
    g.init_state = None
    got = g(x.to(DEV), lens.to(DEV))               # back to normal
    torch.cuda.synchronize()
    assert lib.empose_async_status() == 0
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), atol=1e-4)
    g.release()


@pytest.mark.provokes_poll_timeout
def test_poll_timeout_of_the_large_batch_whole_sequence_kernel_is_reported():
    """The opt-in cooperative kernel for batches above 256 rows (lstm_seq) shares the counter."""
    lib = _lib.lib()
    B, F, In, H, L = 300, 12, 64, 128, 2
    g, sd = _layer(In, H, L, 12)
    x = torch.randn(B, F, In)
    lens = torch.full((B,), F, dtype=torch.int64)
    with torch.no_grad():
        want, _ = R.lstm_forward(sd, 'lstm.', x, lens, None, L, False)
    _lib.check(lib.empose_set_option(b'lstm_seq', 1))
    seen = 0
    for attempt in range(6):
        _lib.check(lib.empose_set_option(b'spin_limit', 1))
        g.init_state = None
        got = g(x.to(DEV), lens.to(DEV))
        torch.cuda.synchronize()
        _lib.check(lib.empose_set_option(b'spin_limit', 0))
        status = lib.empose_async_status()
        has_nan = bool(torch.isnan(got).any()) or bool(torch.isnan(g.final_state[0]).any())
        assert (status == ETIMEOUT) == has_nan
        seen += status == ETIMEOUT
    g.init_state = None
    got = g(x.to(DEV), lens.to(DEV))
    torch.cuda.synchronize()
    assert lib.empose_async_status() == 0
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), atol=1e-4)
    g.release()
    if seen == 0:
        pytest.skip('no poll of lstm_seq gave up with spin_limit = 1 on this run (its waits are short); status path '
                    'covered by the lstm_persist test')


@pytest.mark.provokes_poll_timeout
def test_evaluation_driver_raises_instead_of_averaging_nan():
    """em_pose_amd.eval.helpers._check_async: what evaluate_sequences / evaluate_sequences_batched call once the device is
    in sync."""
    from em_pose_amd.eval.helpers import _check_async
    lib = _lib.lib()
    B, F, In, H, L = 2, 64, 32, 64, 2
    g, _ = _layer(In, H, L, 13)
    x = torch.randn(B, F, In)
    lens = torch.full((B,), F, dtype=torch.int64)
    _check_async(DEV)
    _lib.check(lib.empose_set_option(b'spin_limit', 1))
    raised = False
    for attempt in range(8):
        g.init_state = None
        got = g(x.to(DEV), lens.to(DEV))
        torch.cuda.synchronize()
        try:
            _check_async(DEV)                    # what the drivers do once the device is in sync
            assert not torch.isnan(got).any()    # no report, no NaN
        except _lib.EmposeError as e:
            assert 'timed out' in str(e)
            raised = True
            break
    _lib.check(lib.empose_set_option(b'spin_limit', 0))
    lib.empose_async_status()
    g.release()
    assert raised


@pytest.mark.parametrize('rnn', [True, False], ids=['lgd_rnn', 'lgd'])
def test_training_step_on_side_streams_equals_the_single_stream_step(rnn):
    """nn/train_engine.py with `two_streams` (from 2048 frames per step on: the shape network, the pose network's backward
    and the weight-gradient products on side streams): same kernels, same order per accumulator -- losses, outputs and
    every parameter gradient are bit-identical to the single-stream step.  64 windows x 32 frames = 2048 rows, so the
    BatchNorm statistics also take the GEMM-epilogue + finish-kernel route (train_epi, above 1024 rows)."""
    from em_pose_amd import synthetic
    from em_pose_amd.bodymodels.smpl import SMPLLayer
    from em_pose_amd.data.data import SyntheticBatch
    from em_pose_amd.helpers.configuration import lgd_config
    from em_pose_amd.nn.models import create_model
    from em_pose_amd.nn.train_engine import LgdTrainEngine
    from tests import helpers as H
    model = H.small_model()
    bm = R.BodyModelTensors(model)
    vids = [int(v) for v in np.random.default_rng(5).choice(model['v_template'].shape[0], 12, replace=False)]
    tables = R.sensor_tables(model['f'], vids)

    def sensors(poses, betas, o_r, o_t):
        with torch.no_grad():
            p, o, _ = R.estimated_markers(bm, tables, vids, torch.from_numpy(poses), torch.from_numpy(betas),
                                          torch.from_numpy(o_r), torch.from_numpy(o_t))
        return p.numpy(), o.numpy()
    B, F = 64, 32
    w = synthetic.make_windows(B, F, 3, sensors)
    torch.manual_seed(7)
    net = create_model(lgd_config(12, rnn, 2, hidden=64, rnn_hidden=64), SMPLLayer(model))
    net.vertex_ids = vids
    net = net.to(DEV).train()
    state0 = {k: v.clone() for k, v in net.state_dict().items()}
    lens = torch.full((B,), F, dtype=torch.int64, device=DEV)
    lens[3] = 17
    with torch.no_grad():
        _, _, jgt = R.estimated_markers(bm, tables, vids, torch.from_numpy(w['poses'].reshape(-1, 66)),
                                        torch.from_numpy(np.repeat(w['shapes'], F, axis=0)),
                                        torch.from_numpy(np.repeat(w['offset_r'], F, axis=0)),
                                        torch.from_numpy(np.repeat(w['offset_t'], F, axis=0)))
    res = {}
    try:
        for two in (False, True):
            LgdTrainEngine.two_streams = two
            net.load_state_dict(state0)
            batch = SyntheticBatch(w, lens, device=DEV)
            batch.joints_gt = jgt.reshape(B, F, -1).to(DEV).float()
            net.zero_grad()
            out = net(batch)
            assert net._engine is not None and net._engine._use_side == two
            total, vals = net.backward(batch, out)
            torch.cuda.synchronize()
            res[two] = (vals, {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None},
                        {k: v.detach().clone() for k, v in out.items()},
                        {k: v.clone() for k, v in net.state_dict().items() if 'running' in k})
    finally:
        LgdTrainEngine.two_streams = True
    assert res[True][0] == res[False][0]
    assert res[True][1].keys() == res[False][1].keys() and len(res[True][1]) > 20
    for k, v in res[False][1].items():
        assert torch.equal(res[True][1][k], v), k
    for k, v in res[False][2].items():
        assert torch.equal(res[True][2][k], v), k
    for k, v in res[False][3].items():
        assert torch.equal(res[True][3][k], v), k


def test_full_mesh_split_bf16_with_the_bone_blend_on_the_matrix_cores():
    """Opt-in `mesh_skin_mfma` (mesh_rows_bf16s_kernel: T = sum_b W[v][b] G[f][b] as a second split-bf16 contraction, the
    vector unit only applies it): vertices within 1e-4 m of the oracle and 2e-5 of the fp32 kernel, repeated launches
    bit-identical, frame counts off the 64-frame block, six bones per vertex.  Built for VERDICT r3 item 8, measured 9 %
    slower than the vector skinning and therefore not the default (mesh.hip)."""
    from em_pose_amd import synthetic
    from em_pose_amd.bodymodels.smpl import SMPLLayer
    from tests import helpers as H
    lib = _lib.lib()
    _lib.check(lib.empose_set_option(b'mesh_skin_mfma', 1))
    rng = np.random.default_rng(31)
    big = synthetic.make_model()
    small = dict(H.small_model())
    V = small['v_template'].shape[0]
    w = np.array(small['weights'], dtype=np.float64, copy=True)
    for vtx in range(0, V, 3):
        bones = rng.choice(22, size=6, replace=False)
        w[vtx] = 0
        w[vtx, bones] = rng.uniform(0.1, 1.0, size=6)
        w[vtx] /= w[vtx].sum()
    small['weights'] = w.astype(small['weights'].dtype)
    for model, counts in ((big, (1, 70, 131)), (small, (700,))):
        bm = R.BodyModelTensors(model)
        fast = SMPLLayer(model, arithmetic='bf16x3').to(DEV)
        exact = SMPLLayer(model).to(DEV)
        for n in counts:
            pose = rng.normal(0, 0.5, size=(n, 63)).astype(np.float32)
            root = rng.normal(0, 0.5, size=(n, 3)).astype(np.float32)
            betas = rng.normal(0, 1.5, size=(n, 10)).astype(np.float32)
            trans = rng.normal(0, 1, size=(n, 3)).astype(np.float32)
            v_ref, j_ref = R.smpl_fk(bm, torch.from_numpy(pose), torch.from_numpy(betas), torch.from_numpy(root),
                                     torch.from_numpy(trans))
            kw = dict(poses_body=torch.from_numpy(pose).to(DEV), betas=torch.from_numpy(betas).to(DEV),
                      poses_root=torch.from_numpy(root).to(DEV), trans=torch.from_numpy(trans).to(DEV))
            v, j = fast(**kw)
            v32, j32 = exact(**kw)
            assert float((v.cpu() - v_ref).abs().max()) < 1e-4
            assert float((v - v32).abs().max()) < 2e-5
            assert torch.equal(j, j32)
    g = torch.Generator().manual_seed(5)
    n = 4096
    fast = SMPLLayer(big, arithmetic='bf16x3').to(DEV)
    kw = dict(poses_body=(torch.randn(n, 63, generator=g) * 0.5).to(DEV), betas=torch.randn(n, 10, generator=g).to(DEV),
              poses_root=(torch.randn(n, 3, generator=g) * 0.5).to(DEV))
    first = fast(**kw)[0].clone()
    for _ in range(5):
        assert torch.equal(fast(**kw)[0], first)
    _lib.check(lib.empose_set_option(b'mesh_skin_mfma', 0))
    assert not torch.equal(fast(**kw)[0], first)      # (the option does select another kernel)


def test_weight_gradient_interior_kernel_is_bit_identical_to_the_general_one():
    """gemm_atb_fast_kernel (whole tiles, whole chunks inside one row segment, plain operands) against gemm_atb_lds_kernel:
    same staging, operand order and split, so the same bits -- stand-alone products (with and without accumulation, one
    and many splits) and a training step whose hidden layers take it over the row segments of two applications."""
    lib = _lib.lib()
    g = torch.Generator().manual_seed(3)
    for M, N, K in ((4096, 128, 128), (8192, 256, 384), (1024, 128, 256), (32768, 512, 512)):
        A, B = torch.randn(M, N, generator=g).to(DEV), torch.randn(M, K, generator=g).to(DEV)
        outs = {}
        for fast in (1, 0):
            _lib.check(lib.empose_set_option(b'atb_fast', fast))
            C0 = torch.full((N, K), 0.5, device=DEV)
            bias = torch.full((N,), -0.25, device=DEV)
            nb = lib.empose_gemm_atb_workspace_bytes(M, N, K)
            ws = torch.empty(max(nb, 4), dtype=torch.uint8, device=DEV)
            _lib.check(lib.empose_gemm_atb_f32(M, N, K, A.data_ptr(), N, B.data_ptr(), K, C0.data_ptr(), K, bias.data_ptr(),
                                               ws.data_ptr(), ws.numel(), None))
            torch.cuda.synchronize()
            outs[fast] = (C0.clone(), bias.clone())
        assert torch.equal(outs[1][0], outs[0][0]) and torch.equal(outs[1][1], outs[0][1]), (M, N, K)
        want = A.double().t() @ B.double()
        assert float((outs[1][0].double() - want).abs().max()) < 2e-4 * float(want.abs().max())
        np.testing.assert_allclose(outs[1][1].cpu().numpy(), A.double().sum(0).cpu().numpy(), rtol=0, atol=2e-3)

    # a training step: 2 applications x 2048 rows, hidden 128 -> the hidden layers' dW go through the interior kernel
    from em_pose_amd import synthetic
    from em_pose_amd.bodymodels.smpl import SMPLLayer
    from em_pose_amd.data.data import SyntheticBatch
    from em_pose_amd.helpers.configuration import lgd_config
    from em_pose_amd.nn.models import create_model
    from tests import helpers as H
    model = H.small_model()
    bm = R.BodyModelTensors(model)
    vids = [int(v) for v in np.random.default_rng(5).choice(model['v_template'].shape[0], 12, replace=False)]
    tables = R.sensor_tables(model['f'], vids)

    def sensors(poses, betas, o_r, o_t):
        with torch.no_grad():
            p, o, _ = R.estimated_markers(bm, tables, vids, torch.from_numpy(poses), torch.from_numpy(betas),
                                          torch.from_numpy(o_r), torch.from_numpy(o_t))
        return p.numpy(), o.numpy()
    B_, F = 64, 32
    w = synthetic.make_windows(B_, F, 4, sensors)
    torch.manual_seed(8)
    net = create_model(lgd_config(12, False, 2, hidden=128), SMPLLayer(model))
    net.vertex_ids = vids
    net = net.to(DEV).train()
    state0 = {k: v.clone() for k, v in net.state_dict().items()}
    lens = torch.full((B_,), F, dtype=torch.int64, device=DEV)
    grads = {}
    for fast in (1, 0):
        _lib.check(lib.empose_set_option(b'atb_fast', fast))
        net.load_state_dict(state0)
        batch = SyntheticBatch(w, lens, device=DEV)
        batch.joints_gt = torch.zeros(B_, F, 66, device=DEV)
        net.zero_grad()
        out = net(batch)
        net.backward(batch, out)
        torch.cuda.synchronize()
        grads[fast] = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
    for k, v in grads[0].items():
        assert torch.equal(grads[1][k], v), k
