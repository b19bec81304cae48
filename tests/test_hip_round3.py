"""
GPU parity tests added in round 3: the full-width large-batch LSTM kernel and the
B = 1024 headline batch directly against the oracle, the A^T B workspace boundary (hidden < input width), cache
invalidation after raw-pointer parameter updates, and RCCL (`nccl` backend) initialised on the device with one rank.
Tolerance as everywhere: 1e-4 abs fp32 (BASELINE.json north_star).
"""
import os
import socket

import numpy as np
import pytest
import torch

from em_pose_amd import _lib, synthetic
from em_pose_amd.bodymodels.smpl import SMPLLayer
from em_pose_amd.helpers.configuration import CONSTANTS as CONST
from em_pose_amd.helpers.configuration import lgd_config
from em_pose_amd.nn.layers import RNNLayer
from em_pose_amd.nn.models import create_model
from oracle import torch_ref as R
from tests import helpers as H

pytestmark = pytest.mark.gpu
ATOL = 1e-4
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def big_model():
    return synthetic.make_model()


def _randomize_bn(net, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)


# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('B', [257, 1024])
def test_full_width_lstm_large_batch_vs_oracle(B):
    """`lstm_chain_kernel` at the released width (2 x 512, input 144 = 12 sensors) and more than 256 rows, straight
    against the oracle's explicit LSTM loop: ragged rows, zero and carried initial state, outputs and final state
    (reference layers.py:133-157).  B = 1024 is the headline batch; 257 is the first batch the kernel takes."""
    F, In, Hd, L = 32, 144, 512, 2
    torch.manual_seed(B)
    layer = RNNLayer(In, Hd, L).eval()
    with torch.no_grad():
        for p in layer.lstm.parameters():
            p.mul_(2.0)   # default init is +-1/sqrt(H): make the gates leave their linear range
    x = torch.randn(B, F, In)
    lens = torch.randint(1, F + 1, (B,))
    lens[0], lens[-1] = F, 1
    h0, c0 = 0.5 * torch.randn(L, B, Hd), 0.5 * torch.randn(L, B, Hd)
    sd = {'lstm.' + k: v.detach() for k, v in layer.lstm.state_dict().items()}
    g = layer.to(DEV)
    for state in (None, (h0, c0)):
        with torch.no_grad():
            want, (wh, wc) = R.lstm_forward(sd, 'lstm.', x, lens, state, L, False)
        g.init_state = None if state is None else tuple(t.to(DEV) for t in state)
        got = g(x.to(DEV), lens.to(DEV))
        torch.cuda.synchronize()
        np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), atol=ATOL)
        np.testing.assert_allclose(g.final_state[0].cpu().numpy(), wh.numpy(), atol=ATOL)
        np.testing.assert_allclose(g.final_state[1].cpu().numpy(), wc.numpy(), atol=ATOL)
    g.release()


def test_headline_batch_sampled_windows_vs_oracle(big_model):
    """BASELINE configs[2] at full size (LGD-RNN-12, N = 4, B = 1024 windows of 32 frames, V = 6890): eight windows
    sampled across the batch (first, last, workgroup-tile boundaries) against the oracle run on those windows alone."""
    torch.manual_seed(1615200973)
    net = create_model(lgd_config(12, True, 4), SMPLLayer(big_model))
    _randomize_bn(net, 1615200974)
    net = net.eval()
    bm = R.BodyModelTensors(big_model)
    tables = R.sensor_tables(big_model['f'], CONST.VERTEX_IDS)

    def fn(poses, betas, o_r, o_t):
        with torch.no_grad():
            p, o, _ = R.estimated_markers(bm, tables, CONST.VERTEX_IDS, torch.from_numpy(poses),
                                          torch.from_numpy(betas), torch.from_numpy(o_r), torch.from_numpy(o_t))
        return p.numpy(), o.numpy()
    B, F = 1024, 32
    pool = synthetic.make_windows(64, F, 4242, fn)           # 64 distinct windows, tiled to the batch
    pick = [0, 1, 63, 64, 255, 511, 777, 1023]
    keys = ('marker_pos', 'marker_oris', 'offset_t', 'offset_r')
    order = np.random.default_rng(5).permutation(B) % 64     # window b of the batch is pool window order[b]
    batch = {k: np.ascontiguousarray(pool[k][order]) for k in keys}
    sd = {k: v for k, v in net.state_dict().items() if not k.startswith('smpl.')}
    inp = {k: torch.from_numpy(batch[k][pick]) for k in keys}
    inp['marker_masks'] = None
    inp['seq_lengths'] = torch.full((len(pick),), F, dtype=torch.int64)
    want, _ = R.ief_forward(sd, bm, tables, CONST.VERTEX_IDS, inp, n_markers=12, N=4, rnn_init=True)
    net = net.to(DEV)
    res = net.forward_tensors(*(torch.from_numpy(batch[k]).to(DEV) for k in keys))
    torch.cuda.synchronize()
    pose = res['pose'][pick].cpu().numpy()
    np.testing.assert_allclose(pose[:, :, 3:], want['pose_hat'].numpy(), atol=ATOL)
    np.testing.assert_allclose(pose[:, :, :3], want['root_ori_hat'].numpy(), atol=ATOL)
    np.testing.assert_allclose(res['shape'][pick].cpu().numpy(), want['shape_hat'].numpy(), atol=ATOL)
    np.testing.assert_allclose(res['joints'][pick].cpu().numpy(), want['joints_hat'].numpy(), atol=ATOL)
    dj = (res['joints'][pick].cpu().numpy() - want['joints_hat'].numpy()).reshape(len(pick), F, 22, 3)
    assert np.linalg.norm(dj, axis=-1).mean() * 1000.0 < 0.1   # MPJPE(build, oracle) in mm


# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('rnn', [False, True], ids=['mlp_init', 'rnn_init'])
def test_training_gradients_when_hidden_is_narrower_than_the_input(rnn):
    """the A^T B split workspace must cover EVERY product of a network -- with hidden 256 < input width 296
    (and LSTM hidden 128 < 144 inputs) at >= 4096 rows the (H, H) product needs more split space than the (H, in) one.
    The hand-written step must equal the autograd path over PyTorch ops (which never touches that workspace)."""
    from em_pose_amd.data.data import SyntheticBatch
    model = H.small_model()
    vids = H.load_case('train_lgdrnn12_n2')['meta']['vertex_ids']
    B, F = 136, 32     # 4352 rows
    torch.manual_seed(17)
    net = create_model(lgd_config(12, rnn, 2, hidden=256, rnn_hidden=128), SMPLLayer(model))
    net.vertex_ids = [int(v) for v in vids]
    net = net.to(DEV).train()
    w = synthetic.make_windows(B, F, 23)
    g = torch.Generator().manual_seed(23)
    w['marker_pos'] = torch.randn(B, F, 36, generator=g).numpy()
    w['marker_oris'] = torch.randn(B, F, 108, generator=g).numpy()
    batch = SyntheticBatch(w, device=DEV)
    batch.joints_gt = torch.randn(B, F, 66, generator=g).to(DEV)
    bn_state = {k: v.clone() for k, v in net.state_dict().items() if 'running_' in k or 'num_batches' in k}
    grads, losses = {}, {}
    for engine in (True, False):
        net.use_train_engine = engine
        net.load_state_dict(bn_state, strict=False)
        net.zero_grad()
        out = net(batch)
        assert (net._engine is not None) == engine
        _, losses[engine] = net.backward(batch, out)
        torch.cuda.synchronize()
        grads[engine] = {k: p.grad.detach().cpu().numpy().copy() for k, p in net.named_parameters()
                         if p.grad is not None}
    for k in losses[True]:
        assert losses[True][k] == pytest.approx(losses[False][k], rel=1e-4, abs=1e-6), k
    gmax = max(np.abs(v).max() for v in grads[False].values())
    assert gmax > 0 and set(grads[True]) == set(grads[False]) and len(grads[True]) >= 14
    for k, want in grads[False].items():
        assert np.isfinite(grads[True][k]).all(), k
        np.testing.assert_allclose(grads[True][k], want, atol=3e-3 * max(np.abs(want).max(), 1e-3 * gmax), rtol=3e-3,
                                   err_msg=k)


def test_inference_handle_follows_raw_pointer_updates():
    """HipAdam updates parameters through raw pointers and a replayed HIP graph runs no Python, so neither
    moves a tensor version counter; the cached inference handle (folded BatchNorm, packed weights) must still be rebuilt
    when the model is next used in eval mode."""
    from em_pose_amd.data.data import SyntheticBatch
    from em_pose_amd.helpers.graphed import GraphedTrainStep
    from em_pose_amd.helpers.optim import HipAdam
    model = H.small_model()
    vids = H.load_case('train_lgdrnn12_n2')['meta']['vertex_ids']
    B, F = 4, 8
    torch.manual_seed(5)
    net = create_model(lgd_config(12, True, 2, hidden=32, rnn_hidden=32), SMPLLayer(model))
    net.vertex_ids = [int(v) for v in vids]
    net = net.to(DEV)

    def batch_of(seed):
        w = synthetic.make_windows(B, F, seed)
        g = torch.Generator().manual_seed(seed)
        w['marker_pos'] = torch.randn(B, F, 36, generator=g).numpy()
        w['marker_oris'] = torch.randn(B, F, 108, generator=g).numpy()
        b = SyntheticBatch(w, device=DEV)
        b.joints_gt = torch.randn(B, F, 66, generator=g).to(DEV)
        return b
    probe = batch_of(1)
    args = lambda b: (b.marker_pos_synth, b.marker_ori_synth, b.offset_t_augmented, b.offset_r_augmented)

    net.eval()
    before = net.forward_tensors(*args(probe))['pose'].clone()    # builds and caches the handle
    net.train()
    params = [q for n, q in net.named_parameters() if not n.startswith('smpl.')]
    opt = HipAdam(params, lr=1e-2)
    step = GraphedTrainStep(net, opt, batch_of(2))
    for s in (3, 4, 5):
        step(batch_of(s))
        opt.step()
    torch.cuda.synchronize()
    net.eval()
    cached = net.forward_tensors(*args(probe))['pose'].clone()
    net.release()                                                  # a handle built from scratch
    fresh = net.forward_tensors(*args(probe))['pose'].clone()
    torch.cuda.synchronize()
    assert float((fresh - before).abs().max()) > 1e-4             # three steps at lr 1e-2 moved the outputs
    assert torch.equal(cached, fresh)


def test_hip_adam_skips_missing_gradients_and_restores_state():
    """torch.optim.Adam semantics beyond the plain step: parameters without a gradient are skipped (moments untouched),
    the learning rate is read from `param_groups`, and state written by either optimizer restores into the other."""
    from em_pose_amd.helpers.optim import HipAdam
    torch.manual_seed(4)
    shapes = [(300, 40), (66,), (5000,), (1,)]
    ours = [torch.randn(*s, device=DEV).requires_grad_(True) for s in shapes]
    ref = [p.detach().clone().requires_grad_(True) for p in ours]
    oa, ob = HipAdam(ours, lr=5e-4), torch.optim.Adam(ref, lr=5e-4)

    def one_step(oa, ob, ours, ref, skip):
        for i, (p, q) in enumerate(zip(ours, ref)):
            g = torch.randn_like(p)
            p.grad, q.grad = (None, None) if i == skip else (g.clone(), g.clone())
        oa.step()
        ob.step()
    one_step(oa, ob, ours, ref, skip=None)
    one_step(oa, ob, ours, ref, skip=2)      # parameter 2 sits this step out
    oa.param_groups[0]['lr'] = ob.param_groups[0]['lr'] = 1e-3
    one_step(oa, ob, ours, ref, skip=None)
    torch.cuda.synchronize()
    for p, q in zip(ours, ref):
        np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().cpu().numpy(), atol=2e-6, rtol=1e-5)
    # checkpoint round trip, both directions (all parameters have taken the same number of steps here)
    ours2 = [p.detach().clone().requires_grad_(True) for p in ours]
    ref2 = [p.detach().clone().requires_grad_(True) for p in ours]
    for p in ours + ref + ours2 + ref2:
        p.grad = None
    oa2, ob2 = HipAdam(ours2, lr=1.0), torch.optim.Adam(ref2, lr=1.0)
    a, b = HipAdam(ours, lr=5e-4), torch.optim.Adam(ref, lr=5e-4)
    one_step(a, b, ours, ref, skip=None)
    one_step(a, b, ours, ref, skip=None)
    for p, q in zip(ours2, ours):
        p.data.copy_(q.data)
    for p, q in zip(ref2, ref):
        p.data.copy_(q.data)
    oa2.load_state_dict(b.state_dict())      # torch -> HipAdam
    ob2.load_state_dict(a.state_dict())      # HipAdam -> torch
    assert oa2.lr == 5e-4 and oa2.steps == 2
    one_step(a, b, ours, ref, skip=None)
    gs = [p.grad.clone() for p in ours]
    for p, q, g in zip(ours2, ref2, gs):
        p.grad, q.grad = g.clone(), g.clone()
    oa2.step()
    ob2.step()
    torch.cuda.synchronize()
    for p, q, r_, s_ in zip(ours, ref, ours2, ref2):
        np.testing.assert_allclose(r_.detach().cpu().numpy(), q.detach().cpu().numpy(), atol=2e-6, rtol=1e-5)
        np.testing.assert_allclose(s_.detach().cpu().numpy(), p.detach().cpu().numpy(), atol=2e-6, rtol=1e-5)


def test_pack_inputs_entry_point_equals_indexing():
    """`empose_pack_inputs` (BaseModel.prepare_inputs on GPU tensors) against plain indexing (reference
    models.py:106-125) and the frame weights of loss.py:31-39 x models.py:578-579, 12 and 6 sensors, ragged + masked."""
    from em_pose_amd.nn.models import pack_sensor_inputs
    g = torch.Generator().manual_seed(2)
    B, F = 5, 7
    pos, ori = torch.randn(B, F, 36, generator=g), torch.randn(B, F, 108, generator=g)
    lens = torch.tensor([7, 1, 4, 7, 6])
    masks = (torch.rand(B, F, 12, generator=g) > 0.1).float()
    for subset in (list(range(12)), list(CONST.S_CONFIG_6)):
        x, wgt = pack_sensor_inputs(pos.to(DEV), ori.to(DEV), subset, masks.to(DEV), lens.to(DEV),
                                    want_frame_weight=True)
        want = torch.cat([pos.reshape(B, F, 12, 3)[:, :, subset].reshape(B, F, -1),
                          ori.reshape(B, F, 12, 9)[:, :, subset].reshape(B, F, -1)], dim=-1)
        assert torch.equal(x.cpu(), want)
        live = (torch.arange(F)[None] < lens[:, None]).float() * (F / lens.float())[:, None]
        want_w = live * masks.ne(0).all(-1).float()
        np.testing.assert_allclose(wgt.cpu().numpy(), want_w.reshape(-1).numpy(), rtol=1e-6)
    wide = torch.full((B * F, 80), -1.0, device=DEV)             # into the leading columns of a wider row buffer
    pack_sensor_inputs(pos.to(DEV), ori.to(DEV), list(CONST.S_CONFIG_6), out=wide)
    assert torch.equal(wide[:, :72].cpu(), want.reshape(B * F, 72)) and bool((wide[:, 72:] == -1).all())
    with pytest.raises(_lib.EmposeError):
        _lib.check(_lib.lib().empose_pack_inputs(1, 1, 13, None, None, None, None, None, None, 0, None, None))


# ----------------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_rccl_single_rank_collectives_on_device_tensors():
    """RCCL on the MI355X: the `nccl` backend initialised with world size 1 on the device (`device_id=`), all_gather /
    all_reduce / barrier round trips on device tensors, the package's own users of it -- bucketed gradient averaging
    (`helpers/distributed.py`) and the metric gather (`eval/metrics.py`) -- run through the initialised group."""
    import torch.distributed as dist
    from em_pose_amd.eval.metrics import MetricsEngine
    from em_pose_amd.helpers.distributed import GradientBuckets, allreduce_gradients
    assert not dist.is_initialized()
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(_free_port())
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dev = torch.device(DEV)
    dist.init_process_group(backend='nccl', rank=0, world_size=1, device_id=dev)
    try:
        assert dist.get_backend() == 'nccl'
        x = torch.arange(1 << 20, dtype=torch.float32, device=dev)
        y = x.clone()
        dist.all_reduce(y, op=dist.ReduceOp.SUM)
        parts = [torch.empty_like(x)]
        dist.all_gather(parts, x)
        dist.barrier()
        torch.cuda.synchronize()
        assert torch.equal(y, x) and torch.equal(parts[0], x)
        # gradient averaging over persistent flat buckets, side stream
        ps = [torch.randn(n, device=dev).requires_grad_(True) for n in (70000, 3, 512 * 512, 66)]
        want = []
        for p in ps:
            p.grad = torch.randn_like(p)
            want.append(p.grad.clone())
        buckets = GradientBuckets(ps, bucket_bytes=1 << 20, force=True)
        assert buckets.n_buckets >= 2
        for p in ps:
            buckets.stage(p)
        buckets.finish()
        torch.cuda.synchronize()
        for p, w_ in zip(ps, want):
            assert torch.equal(p.grad, w_)                      # mean over one rank
        assert allreduce_gradients(ps) == 0                     # world size 1: nothing to do
        # the training engine writing straight into the buckets, collectives overlapped with the reverse sweep:
        # same gradients as without a sink, `.grad` IS the bucket slice, same addresses on the next step
        from em_pose_amd.data.data import SyntheticBatch
        from em_pose_amd.helpers.distributed import attach_gradient_buckets
        from em_pose_amd.nn.train_engine import LgdTrainEngine
        torch.manual_seed(5)
        net = create_model(lgd_config(12, True, 2, hidden=32, rnn_hidden=32), SMPLLayer(H.small_model()))
        net.vertex_ids = [int(v) for v in H.load_case('train_lgdrnn12_n2')['meta']['vertex_ids']]
        net = net.to(dev).train()
        w = synthetic.make_windows(4, 8, 3)
        gen = torch.Generator().manual_seed(3)
        w['marker_pos'] = torch.randn(4, 8, 36, generator=gen).numpy()
        w['marker_oris'] = torch.randn(4, 8, 108, generator=gen).numpy()
        batch = SyntheticBatch(w, device=dev)
        batch.joints_gt = torch.randn(4, 8, 66, generator=gen).to(dev)
        bn_state = {k: v.clone() for k, v in net.state_dict().items() if 'running_' in k or 'num_batches' in k}
        own = [q for n, q in net.named_parameters() if not n.startswith('smpl.')]
        net.zero_grad()
        net.backward(batch, net(batch))
        plain = [q.grad.clone() for q in own]
        order = LgdTrainEngine.gradient_order(net)
        assert {id(q) for q in order} == {id(q) for q in own}
        nb = GradientBuckets(order, bucket_bytes=16 << 10, force=True)
        assert nb.n_buckets >= 3
        attach_gradient_buckets(net, nb)
        ptrs = None
        for _ in range(2):
            net.load_state_dict(bn_state, strict=False)
            net.zero_grad()
            net.backward(batch, net(batch))
            assert nb.finish() == nb.n_buckets
            torch.cuda.synchronize()
            for q, want_g in zip(own, plain):
                assert q.grad.data_ptr() == nb.view_of(q).data_ptr()
                assert torch.equal(q.grad, want_g)
            now = [q.grad.data_ptr() for q in own]
            assert ptrs is None or ptrs == now
            ptrs = now
        attach_gradient_buckets(net, None)
        # metric accumulators through the group
        me = MetricsEngine(None)
        j = torch.randn(2, 5, 66, device=dev)
        me.compute_joint_dist(j, j + 0.01)
        rows = me.state()['eucl'].shape[0]
        me.gather(device=dev, force=True)
        assert me.state()['eucl'].shape[0] == rows
    finally:
        dist.destroy_process_group()
