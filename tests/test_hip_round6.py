"""
Round 6 GPU tests (run with `-m gpu` on an MI355X).

  * kernels on the three-piece bf16 MFMA run one workgroup per CU by construction: repeat launches are bit-identical at a
    narrow LSTM width, where their own LDS footprint would have let two or three workgroups share a CU
  * the one-launch training layers beside a long kernel on another stream: correct gradients, no poll timeout (VERDICT r5
    item 8: eager launches are cooperative launches)
"""
import numpy as np
import pytest
import torch

from em_pose_amd import _lib, synthetic
from em_pose_amd.bodymodels.smpl import SMPLLayer
from em_pose_amd.helpers.configuration import CONSTANTS as CONST
from em_pose_amd.helpers.configuration import lgd_config
from em_pose_amd.nn.models import create_model
from oracle import torch_ref as R

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def big_model():
    return synthetic.make_model()


class _Option(object):
    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        self.old = _lib.lib().empose_get_option(self.name)
        _lib.check(_lib.lib().empose_set_option(self.name, self.value))

    def __exit__(self, *exc):
        _lib.lib().empose_set_option(self.name, self.old)
        return False


@pytest.mark.parametrize('rnn_hidden', [128, 256])
def test_three_piece_bf16_row_kernels_repeat_bit_identically_at_narrow_widths(big_model, rnn_hidden):
    """LGD-RNN-12, N = 2, with a 128- / 256-wide LSTM, 160 windows x 32 frames, frame-per-lane SMPL path forced: the init
    heads (heads_rows_kernel<true>, 33 / 67 KB of LDS of its own) and both blend products (blend_feat / blend_t, X3) run on
    v_mfma_f32_32x32x16_bf16.  Two waves of one SIMD issuing that instruction leave sporadic wrong accumulator elements
    (scripts/dev/bf16_hazard_repro.md), so their launches request more than half a CU's LDS: one workgroup per CU whatever
    the width.  Six forwards, the same bits every time, and right (eight sampled windows against the oracle)."""
    torch.manual_seed(rnn_hidden)
    net = create_model(lgd_config(12, True, 2, rnn_hidden=rnn_hidden), SMPLLayer(big_model)).eval()
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
    bm = R.BodyModelTensors(big_model)
    tables = R.sensor_tables(big_model['f'], CONST.VERTEX_IDS)

    def fn(poses, betas, o_r, o_t):
        with torch.no_grad():
            p, o, _ = R.estimated_markers(bm, tables, CONST.VERTEX_IDS, torch.from_numpy(poses),
                                          torch.from_numpy(betas), torch.from_numpy(o_r), torch.from_numpy(o_t))
        return p.numpy(), o.numpy()
    B, F = 160, 32
    pool = synthetic.make_windows(32, F, 99, fn)
    order = np.random.default_rng(3).permutation(B) % 32
    keys = ('marker_pos', 'marker_oris', 'offset_t', 'offset_r')
    batch = {k: np.ascontiguousarray(pool[k][order]) for k in keys}
    pick = [0, 1, 63, 64, 65, 127, 128, B - 1]
    sd = {k: v for k, v in net.state_dict().items() if not k.startswith('smpl.')}
    inp = {k: torch.from_numpy(batch[k][pick]) for k in keys}
    inp['marker_masks'] = None
    inp['seq_lengths'] = torch.full((len(pick),), F, dtype=torch.int64)
    want, _ = R.ief_forward(sd, bm, tables, CONST.VERTEX_IDS, inp, n_markers=12, N=2, rnn_init=True)
    net = net.to(DEV)
    dev_in = [torch.from_numpy(batch[k]).to(DEV) for k in keys]
    runs = []
    with _Option(b'smpl_tile', 2), _Option(b'rows_x3', 1):
        for _ in range(6):
            res = net.forward_tensors(*dev_in)
            torch.cuda.synchronize()
            runs.append({k: res[k].cpu().numpy().copy() for k in ('pose', 'shape', 'joints')})
    for r in runs[1:]:
        for k in r:
            assert np.array_equal(r[k], runs[0][k]), k
    pose = runs[0]['pose'][pick]
    np.testing.assert_allclose(pose[:, :, 3:], want['pose_hat'].numpy(), atol=1e-4)
    np.testing.assert_allclose(pose[:, :, :3], want['root_ori_hat'].numpy(), atol=1e-4)
    np.testing.assert_allclose(runs[0]['shape'][pick], want['shape_hat'].numpy(), atol=1e-4)
    np.testing.assert_allclose(runs[0]['joints'][pick], want['joints_hat'].numpy(), atol=1e-4)


def test_one_launch_training_layers_beside_a_long_kernel_on_another_stream():
    """The workgroups of a one-launch training layer (csrc/train_cols.hip) wait for each other's exchange words: all of
    them must be resident.  Eager launches are cooperative launches (hipLaunchCooperativeKernel), so the runtime -- not an
    occupancy argument -- guarantees it, whatever else is on the device.  Here a second stream keeps every CU busy with long
    element-wise kernels while the paired layers of a 12-window step (384 rows, both update networks) run forward and
    backward: the results are those of the same call alone on the device, bit for bit, and no poll gave up."""
    from tests.test_hip_round5 import _mlp_pair, _run_mlp_train
    lib = _lib.lib()
    assert lib.empose_get_option(b'cols_coop') == 1 and lib.empose_get_option(b'train_cols') == 1
    M, in_dim, hidden = 384, 296, 512
    g = torch.Generator().manual_seed(3)
    x = torch.randn(M, in_dim, generator=g).to(DEV)
    d_outs = [torch.zeros(M, 68, device=DEV), torch.zeros(M, 12, device=DEV)]
    d_outs[0][:, :66] = torch.randn(M, 66, generator=g).to(DEV)
    d_outs[1][:, :10] = torch.randn(M, 10, generator=g).to(DEV)
    assert lib.empose_async_status() == 0
    alone = _run_mlp_train(_mlp_pair(in_dim, hidden, 5), x, d_outs, M, pair=True)
    side = torch.cuda.Stream()
    big = torch.randn(1 << 28, device=DEV)          # 1 GiB: each pass over it occupies the whole chip for ~0.3 ms
    torch.cuda.synchronize()
    for rep in range(3):
        done = torch.cuda.Event()
        with torch.cuda.stream(side):
            for _ in range(60):
                big.mul_(1.0000001).add_(1e-9)
            done.record()
        nets = _mlp_pair(in_dim, hidden, 5)
        res = _run_mlp_train(nets, x, d_outs, M, pair=True)      # (synchronises and checks empose_async_status)
        for a, b in zip(res['out'] + res['grads'][0] + res['grads'][1], alone['out'] + alone['grads'][0] + alone['grads'][1]):
            assert torch.isfinite(a).all()
            assert torch.equal(a, b)
        side.synchronize()
    assert lib.empose_async_status() == 0


@pytest.mark.parametrize('n', [200, 4096 + 17, 16384])
def test_full_mesh_three_piece_bf16_is_as_accurate_as_the_fp32_mfma_kernel(big_model, n):
    """mesh_x3.hip (round 6, the default full-mesh path): the blend-shape contraction of all 200 columns as six bf16 MFMA
    products of three bf16 pieces per operand, fp32 accumulate.  Claim: fp32-EQUIVALENT -- against a float64 evaluation of
    the same body model (V = 6890: 215 whole vertex tiles + one of 10 vertices; frame counts off the 64-frame block) its
    error is no larger than that of the kernel on the fp32 MFMA instruction (option mesh_x3 = 0) -- with a tile's stores
    staggered into the next tile's K loop (1, the default), at the end of their tile (2), and with the skinning
    software-pipelined under the next tile's products (3): the three agree bit for bit; repeated launches are bit-identical
    (one wave per SIMD, and no store while the wave has MFMAs in flight); joints do not depend on the option."""
    lib = _lib.lib()
    rng = np.random.default_rng(n)
    pose = rng.normal(0, 0.5, size=(n, 63)).astype(np.float32)
    root = rng.normal(0, 0.5, size=(n, 3)).astype(np.float32)
    betas = rng.normal(0, 1.5, size=(n, 10)).astype(np.float32)
    trans = rng.normal(0, 1, size=(n, 3)).astype(np.float32)
    bm64 = R.BodyModelTensors(big_model, dtype=torch.float64)
    pick = np.unique(np.concatenate([np.arange(0, n, max(1, n // 96)), [n - 1, n - 2, 63, 64, 65]]))
    pick = pick[pick < n]
    v64, j64 = R.smpl_fk(bm64, torch.from_numpy(pose[pick]).double(), torch.from_numpy(betas[pick]).double(),
                         torch.from_numpy(root[pick]).double(), torch.from_numpy(trans[pick]).double())
    smpl = SMPLLayer(big_model).to(DEV)
    kw = dict(poses_body=torch.from_numpy(pose).to(DEV), betas=torch.from_numpy(betas).to(DEV),
              poses_root=torch.from_numpy(root).to(DEV), trans=torch.from_numpy(trans).to(DEV))
    err, out = {}, {}
    for opt in (0, 1, 2, 3):
        with _Option(b'mesh_x3', opt):
            v, j = smpl(**kw)
            torch.cuda.synchronize()
            for _ in range(3 if opt else 1):
                v2, _ = smpl(**kw)
                assert torch.equal(v, v2)
        out[opt] = (v.cpu(), j.cpu())
        err[opt] = float((out[opt][0][pick].double() - v64).abs().max())
    print('full mesh, %d frames vs float64: fp32 MFMA %.2e, three-piece bf16 %.2e (staggered stores) %.2e (stores at the end '
          'of the tile) %.2e (skinning under the products)' % (n, err[0], err[1], err[2], err[3]))
    assert torch.equal(out[1][0], out[2][0]) and torch.equal(out[1][0], out[3][0])
    assert all(torch.equal(out[0][1], out[o][1]) for o in (1, 2, 3))
    assert err[1] <= 1.5 * err[0] + 1e-7 and err[1] < 2e-5
    assert float((out[1][0] - out[0][0]).abs().max()) < 1e-5


def test_full_mesh_three_piece_bf16_on_a_small_mesh_and_few_frames():
    """The same kernel where the grid splits the mesh's tiles over workgroups (few frames), on the 160-vertex model, with
    and without a translation, against the oracle (2e-5, the bar of test_full_mesh_vertices_vs_oracle)."""
    from tests import helpers as H
    model = H.small_model()
    bm = R.BodyModelTensors(model)
    smpl = SMPLLayer(model).to(DEV)
    rng = np.random.default_rng(4)
    for n, with_trans in ((1, True), (3, False), (65, True), (700, False)):
        pose = rng.normal(0, 0.4, size=(n, 63)).astype(np.float32)
        root = rng.normal(0, 0.5, size=(n, 3)).astype(np.float32)
        betas = rng.normal(0, 1, size=(n, 10)).astype(np.float32)
        trans = rng.normal(0, 1, size=(n, 3)).astype(np.float32) if with_trans else None
        v_ref, j_ref = R.smpl_fk(bm, torch.from_numpy(pose), torch.from_numpy(betas), torch.from_numpy(root),
                                 torch.from_numpy(trans) if with_trans else None)
        for opt in (1, 2, 3):
            with _Option(b'mesh_x3', opt):
                v, j = smpl(poses_body=torch.from_numpy(pose).to(DEV), betas=torch.from_numpy(betas).to(DEV),
                            poses_root=torch.from_numpy(root).to(DEV),
                            trans=torch.from_numpy(trans).to(DEV) if with_trans else None)
            np.testing.assert_allclose(v.cpu().numpy(), v_ref.numpy(), atol=2e-5)
            np.testing.assert_allclose(j.cpu().numpy(), j_ref.numpy(), atol=2e-5)


def _dense_names(mlp):
    """(linear, batch norm, activation) state_dict prefixes of an MLP's layers in order (reference nn/layers.py:13-77)."""
    mods = dict(mlp.named_modules())
    names = {id(m): n for n, m in mods.items()}
    return [(names[id(lin)], None if bn is None else names[id(bn)], None if act is None else names[id(act)])
            for lin, bn, act in mlp.dense_specs()]


@pytest.mark.parametrize('M,in_dim', [(8192, 296), (2048 + 40, 224)])
def test_training_layer_products_on_three_bf16_pieces(M, in_dim):
    """Large training batches (round 6, gemm_train_x3_kernel): the forward products y = a W^T (statistics epilogue) and the
    reverse products dA = dY W (dyh epilogue) of the two update networks on three bf16 pieces per operand, the weights packed
    once per step (empose_pack_weight_x3), against the same calls on the fp32 MFMA tile (option train_x3 = 0) -- outputs,
    saved activations, BatchNorm statistics, cotangent stashes, BatchNorm / PReLU gradients, weight gradients -- to rounding
    (the reverse sweeps read ONE set of saved activations, so no PReLU branch differs), and the forward against a float64
    evaluation: no less accurate than the fp32 instruction.  Repeated calls are bit-identical."""
    from tests.test_hip_round5 import _mlp_pair, _run_mlp_train
    lib = _lib.lib()
    hidden = 512
    g = torch.Generator().manual_seed(M)
    x = torch.randn(M, in_dim, generator=g).to(DEV)
    d_outs = [torch.zeros(M, 68, device=DEV), torch.zeros(M, 12, device=DEV)]
    d_outs[0][:, :66] = torch.randn(M, 66, generator=g).to(DEV)
    d_outs[1][:, :10] = torch.randn(M, 10, generator=g).to(DEV)
    res = {}
    for key, opt in (('fp32', 0), ('x3', 1), ('x3_again', 1)):
        nets = _mlp_pair(in_dim, hidden, 5)
        with _Option(b'train_x3', opt):
            res[key] = _run_mlp_train(nets, x, d_outs, M, pair=False, deferred=False,
                                      saves_from=None if key == 'fp32' else res['fp32']['save'])
    def flat(r):
        return list(r['out']) + list(r['save']) + [t for b in r['bn'] for t in b] + [t for gl in r['grads'] for t in gl]
    for a, b in zip(flat(res['x3']), flat(res['x3_again'])):
        assert torch.equal(a, b)
    worst = 0.0
    for idx, (a, b) in enumerate(zip(flat(res['fp32']), flat(res['x3']))):
        assert torch.isfinite(b.float()).all()
        scale = max(1.0, float(a.abs().max()))
        err = float((a.double() - b.double()).abs().max()) / scale
        worst = max(worst, err)
        assert err < (1e-4 if a.ndim == 1 and a.numel() == hidden else 2e-5), (err, idx, a.shape)
    # the forward against float64 (train-mode BatchNorm over the batch, reference nn/layers.py:13-77)
    errs = {}
    for key in ('fp32', 'x3'):
        worst64 = 0.0
        for i, net in enumerate(_mlp_pair(in_dim, hidden, 5)):
            sd = {k: v.detach().double() for k, v in net.state_dict().items()}
            h = x.double()
            for lin, bn, act in [(n_[0], n_[1], n_[2]) for n_ in _dense_names(net)]:
                h = h @ sd[lin + '.weight'].T + sd[lin + '.bias']
                if bn is not None:
                    mean, var = h.mean(0), h.var(0, unbiased=False)
                    h = (h - mean) / torch.sqrt(var + 1e-5) * sd[bn + '.weight'] + sd[bn + '.bias']
                    h = torch.where(h > 0, h, sd[act + '.weight'] * h)
            want = h
            worst64 = max(worst64, float((res[key]['out'][i].double() - want).abs().max()) / max(1.0, float(want.abs().max())))
        errs[key] = worst64
    print('training layers M=%d in=%d: three pieces vs fp32 tile worst relative difference %.2e; forward vs float64: fp32 %.2e, '
          'three pieces %.2e' % (M, in_dim, worst, errs['fp32'], errs['x3']))
    assert errs['x3'] <= 1.5 * errs['fp32'] + 1e-7


def test_three_piece_kernels_keep_their_bits_beside_a_storing_kernel_on_another_stream(big_model):
    """A global store issued by ANY wave of a SIMD while v_mfma_f32_32x32x16_bf16 is in flight there corrupts an accumulator
    element (scripts/dev/bf16_hazard_repro.md).  Every kernel built on that instruction therefore allocates all 512 registers
    of its SIMD lane (X3_EXCLUSIVE_SIMD, csrc/bf16x3.h): no wave of any other kernel -- another stream of this process, as
    the training engine's side streams and the streaming evaluation driver use them, or another process -- fits beside it.
    Here a second stream runs element-wise kernels (stores, no LDS, few registers: they fit anywhere there is room) without
    pause while the headline-shaped forward (update MLPs, LSTM steps, heads, blend GEMMs on three pieces) and the full-mesh
    evaluation repeat on the first one: every repetition has the bits of the run alone on the device.  A guard, not the
    reproducer: found in round 6 when the three-piece training GEMMs met the engine's side streams -- two runs of the same
    64-window step differed in the eighth digit; `scripts/dev/x3_shared_simd_lab.sh` shows five runs / five results without
    the macro and five / one with it, and tests/test_bench_contract.py::test_training_with_buckets_rccl_and_side_streams_
    equals_the_plain_step holds the exact equality in the suite."""
    torch.manual_seed(3)
    net = create_model(lgd_config(12, True, 4), SMPLLayer(big_model)).eval().to(DEV)
    B, F = 512, 32
    g = torch.Generator().manual_seed(7)
    inputs = [torch.randn(B, F, 36, generator=g).to(DEV), torch.randn(B, F, 108, generator=g).to(DEV),
              (0.02 * torch.randn(B, 12, 3, generator=g)).to(DEV),
              torch.eye(3).repeat(B, 12, 1, 1).to(DEV)]
    smpl = SMPLLayer(big_model).to(DEV)
    T = 4096
    kw = dict(poses_body=(torch.randn(T, 63, generator=g) * 0.5).to(DEV), betas=torch.randn(T, 10, generator=g).to(DEV),
              poses_root=(torch.randn(T, 3, generator=g) * 0.5).to(DEV))

    def run():
        res = net.forward_tensors(*inputs)
        v, j = smpl(**kw)
        torch.cuda.synchronize()
        return [res[k].clone() for k in ('pose', 'shape', 'joints')] + [v.clone()]
    alone = run()
    assert all(torch.isfinite(t).all() for t in alone)
    side = torch.cuda.Stream()
    big = torch.randn(1 << 26, device=DEV)
    stop = []
    for rep in range(4):
        with torch.cuda.stream(side):
            for _ in range(400):
                big.mul_(1.0000001).add_(1e-9)
        got = run()
        for a, b in zip(got, alone):
            assert torch.equal(a, b)
        side.synchronize()


# ----------------------------------------------------------------------------------------------------------------------
# Medium batches, the whole sequence in one cooperative launch (csrc/lstm_midseq_x3.hip)
@pytest.mark.parametrize('B,F,In,Hd,L', [(36, 64, 72, 512, 2), (17, 9, 144, 512, 2), (32, 5, 144, 256, 2), (64, 33, 72, 64, 1),
                                         (20, 16, 36, 128, 4), (33, 4, 72, 192, 3),
                                         (4, 12, 72, 512, 2), (12, 40, 72, 512, 2), (16, 7, 144, 192, 3)])
def test_medium_batch_whole_sequence_lstm_equals_its_step_launches(B, F, In, Hd, L):
    """lstm_midseq_x3_kernel (4 .. 64 rows: weights in registers for all steps, hidden states handed over through fresh
    A planes and progress counters; opt-in, option lstm_midseq = 1) against the launches per wavefront step: from 17 rows on those
    are lstm_mid_x3_kernel -- same tiles, same products in the same order -- and every output has the same BITS; below, the
    step launches are the fp32 kernels of a few rows and both are held to a float64 LSTM (reference nn/layers.py:133-157:
    ragged rows, carried state, zero-padded outputs).  Repeated runs are bit-identical."""
    from em_pose_amd.nn.layers import RNNLayer
    torch.manual_seed(B * 7 + F)
    layer = RNNLayer(In, Hd, L).eval()
    with torch.no_grad():
        for p in layer.lstm.parameters():
            p.mul_(2.0)
    x = torch.randn(B, F, In)
    lens = torch.randint(1, F + 1, (B,))
    lens[0], lens[-1] = F, 1
    h0, c0 = 0.5 * torch.randn(L, B, Hd), 0.5 * torch.randn(L, B, Hd)
    sd64 = {'lstm.' + k: v.detach().double() for k, v in layer.lstm.state_dict().items()}
    with torch.no_grad():
        want = {st is not None: R.lstm_forward(sd64, 'lstm.', x.double(), lens, None if st is None else (h0.double(), c0.double()),
                                               L, False) for st in (None, 1)}
    g = layer.to(DEV)
    outs, err = {}, {}
    for one_launch in (0, 1):
        # (the step launches it shares its bits with: the 8-unit tiles of lstm_mid_x3.hip, not the 4-unit ones that are the default)
        with _Option(b'lstm_midseq', one_launch), _Option(b'lstm_mid16', 0):
            first, worst = None, 0.0
            for rep in range(3 if one_launch else 1):
                got_all = []
                for carried in (False, True):
                    g.init_state = (h0.to(DEV), c0.to(DEV)) if carried else None
                    y = g(x.to(DEV), lens.to(DEV))
                    torch.cuda.synchronize()
                    wy, (wh, wc) = want[carried]
                    got = (y.cpu(), g.final_state[0].cpu(), g.final_state[1].cpu())
                    got_all += [t.numpy() for t in got]
                    worst = max(worst, float((got[0].double() - wy).abs().max()), float((got[1].double() - wh).abs().max()),
                                float((got[2].double() - wc).abs().max()))
                if first is None:
                    first = got_all
                else:
                    for a_, b_ in zip(first, got_all):
                        assert np.array_equal(a_, b_)
            outs[one_launch], err[one_launch] = first, worst
    assert _lib.lib().empose_async_status() == 0
    print('lstm %s vs float64: step launches %.2e, one launch %.2e' % ((B, F, In, Hd, L), err[0], err[1]))
    assert err[1] < 1e-5 and err[1] <= 1.5 * err[0] + 2e-7
    if B >= 17:
        for a_, b_ in zip(outs[0], outs[1]):
            assert np.array_equal(a_, b_)
    g.release()


@pytest.mark.provokes_poll_timeout
def test_medium_batch_whole_sequence_lstm_reports_a_poll_that_gave_up():
    """spin_limit = 1: the polls of lstm_midseq_x3_kernel give up almost at once; empose_async_status() says
    EMPOSE_ETIMEOUT exactly when the outputs hold NaN, and with the normal limit the layer gives the reference's numbers."""
    from em_pose_amd.nn.layers import RNNLayer
    lib = _lib.lib()
    assert lib.empose_async_status() == 0
    B, F, In, H, L = 24, 48, 72, 512, 2
    torch.manual_seed(3)
    layer = RNNLayer(In, H, L).eval()
    sd = {'lstm.' + k: v.detach().clone() for k, v in layer.lstm.state_dict().items()}
    g = layer.to(DEV)
    x = torch.randn(B, F, In)
    lens = torch.full((B,), F, dtype=torch.int64)
    with torch.no_grad():
        want, _ = R.lstm_forward(sd, 'lstm.', x, lens, None, L, False)
    timed_out = 0
    _lib.check(lib.empose_set_option(b'lstm_midseq', 1))
    g.release()                       # (the workspace is carved for the kernels the options select)
    for attempt in range(12):
        if timed_out >= 2:
            break
        _lib.check(lib.empose_set_option(b'spin_limit', 1))
        g.init_state = None
        got = g(x.to(DEV), lens.to(DEV))
        torch.cuda.synchronize()
        _lib.check(lib.empose_set_option(b'spin_limit', 0))
        has_nan = bool(torch.isnan(got).any()) or bool(torch.isnan(g.final_state[0]).any())
        status = lib.empose_async_status()
        assert (status == -4) == has_nan, (status, has_nan)
        if status == -4:
            timed_out += 1
            assert b'timed out' in lib.empose_last_error()
        else:
            np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), atol=1e-4)
    assert timed_out >= 1, 'spin_limit = 1 never made a poll give up'
    g.init_state = None
    got = g(x.to(DEV), lens.to(DEV))
    torch.cuda.synchronize()
    assert lib.empose_async_status() == 0
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), atol=1e-4)
    g.release()


# ----------------------------------------------------------------------------------------------------------------------
# Dropout flags of the LGD model (reference configuration.py:164,172; layers.py:30,62,103-106,140)
def test_lgd_training_with_dropout_flags_runs_on_the_autograd_path():
    """`m_dropout` (inputs of the init RNN) and `m_dropout_hidden` (the update MLPs) > 0: no released configuration sets
    them, the reference accepts them.  The hand-written training engine declines, the autograd path over the same kernels
    takes the step: masks are drawn (two training-mode forwards differ), every parameter gets a finite gradient, and eval
    mode is the network without dropout, bit for bit."""
    from em_pose_amd.data.data import SyntheticBatch
    from tests import helpers as H
    case = H.load_case('train_lgdrnn12_n2')
    meta, w = case['meta'], case['in']
    vids = [int(v) for v in meta['vertex_ids']]
    nets = []
    for p_in, p_hid in ((0.2, 0.3), (0.0, 0.0)):
        cfg = lgd_config(12, True, 2, hidden=32, rnn_hidden=32, m_dropout=p_in, m_dropout_hidden=p_hid)
        net = create_model(cfg, SMPLLayer(H.small_model()))
        missing, unexpected = net.load_state_dict(H.sd_to_torch(case['sd']), strict=False)
        assert not unexpected and all(k.startswith('smpl.') for k in missing)
        net.vertex_ids = vids
        nets.append(net.to(DEV))
    net, plain = nets
    batch = SyntheticBatch(w, torch.from_numpy(w['seq_lengths']).to(DEV), device=DEV)
    batch.joints_gt = torch.from_numpy(w['joints_gt']).to(DEV)
    net.eval(); plain.eval()
    a, b = net(batch), plain(batch)
    for k in ('pose_hat', 'shape_hat', 'root_ori_hat'):
        assert torch.equal(a[k], b[k]), k
    net.train()
    net.zero_grad()
    o1 = net(batch)
    assert net._engine is None       # (nn/train_engine.py covers dropout 0 only)
    p1 = o1['pose_hat'].detach().clone()
    total, vals = net.backward(batch, o1)
    o2 = net(batch)
    assert not torch.equal(p1, o2['pose_hat'].detach())
    assert np.isfinite(vals['total_loss'])
    n = 0
    for name, p in net.named_parameters():
        if name.startswith('smpl.') or not p.requires_grad:
            continue
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), name
        n += 1
    assert n >= 14


@pytest.mark.parametrize('B,F,In,Hd,L', [(36, 9, 72, 512, 2), (17, 5, 144, 512, 2), (32, 4, 60, 256, 2), (64, 6, 72, 64, 1),
                                         (33, 3, 36, 128, 4), (20, 5, 100, 96, 3), (49, 7, 72, 512, 2)])
def test_lstm_steps_on_sixteen_column_tiles_are_as_accurate_as_the_thirty_two_column_ones(B, F, In, Hd, L):
    """lstm_mid16_x3_kernel (17 .. 64 rows: 4 hidden units x 4 gates per workgroup on all 256 CUs, v_mfma_f32_16x16x32_bf16,
    k-steps of 32 -- an input whose k-steps of 16 are odd in number ends on a half step) against a float64 LSTM (reference
    nn/layers.py:133-157: ragged rows, carried state, zero-padded outputs): no less accurate than lstm_mid_x3_kernel, and
    repeated runs are bit-identical."""
    from em_pose_amd.nn.layers import RNNLayer
    torch.manual_seed(B * 5 + F)
    layer = RNNLayer(In, Hd, L).eval()
    with torch.no_grad():
        for p in layer.lstm.parameters():
            p.mul_(2.0)
    x = torch.randn(B, F, In)
    lens = torch.randint(1, F + 1, (B,))
    lens[0], lens[-1] = F, 1
    h0, c0 = 0.5 * torch.randn(L, B, Hd), 0.5 * torch.randn(L, B, Hd)
    sd64 = {'lstm.' + k: v.detach().double() for k, v in layer.lstm.state_dict().items()}
    with torch.no_grad():
        want = {st is not None: R.lstm_forward(sd64, 'lstm.', x.double(), lens, None if st is None else (h0.double(), c0.double()),
                                               L, False) for st in (None, 1)}
    g = layer.to(DEV)
    err = {}
    for tiles16 in (0, 1):
        with _Option(b'lstm_mid16', tiles16):
            first, worst = None, 0.0
            for rep in range(3 if tiles16 else 1):
                got_all = []
                for carried in (False, True):
                    g.init_state = (h0.to(DEV), c0.to(DEV)) if carried else None
                    y = g(x.to(DEV), lens.to(DEV))
                    torch.cuda.synchronize()
                    wy, (wh, wc) = want[carried]
                    got = (y.cpu(), g.final_state[0].cpu(), g.final_state[1].cpu())
                    got_all += [t.numpy() for t in got]
                    worst = max(worst, float((got[0].double() - wy).abs().max()), float((got[1].double() - wh).abs().max()),
                                float((got[2].double() - wc).abs().max()))
                if first is None:
                    first = got_all
                else:
                    for a_, b_ in zip(first, got_all):
                        assert np.array_equal(a_, b_)
            err[tiles16] = worst
    print('lstm %s vs float64: 32-column tiles %.2e, 16-column tiles %.2e' % ((B, F, In, Hd, L), err[0], err[1]))
    assert err[1] < 1e-5 and err[1] <= 1.5 * err[0] + 2e-7
    g.release()
