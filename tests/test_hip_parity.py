"""
GPU parity tests (run with `-m gpu` on an MI355X): every HIP entry point of include/empose_hip.h against the oracle
(oracle/torch_ref.py pinned to the reference; oracle/analytic_np.py for per-kernel intermediates) and against the
golden vectors recorded from the reference itself.

Tolerance: BASELINE.json north_star -- outputs (pose, shape, joints, vertices) within 1e-4 abs fp32 of the reference
PyTorch-CPU path. Gradient features are O(10) so they carry a matching relative tolerance.
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from em_pose_amd import _lib, synthetic
from em_pose_amd.bodymodels import tables as TB
from em_pose_amd.bodymodels.smpl import SMPLLayer
from em_pose_amd.helpers.configuration import CONSTANTS as CONST
from em_pose_amd.helpers.configuration import lgd_config
from em_pose_amd.nn.models import create_model
from oracle import analytic_np as A
from oracle import torch_ref as R
from tests import helpers as H

pytestmark = pytest.mark.gpu
ATOL = 1e-4
DEV = 'cuda:0'


def _set_option(name, value):
    """Selects a kernel variant for the rest of THIS test: tests/conftest.py's autouse fixture puts every option back to
    its default (empose_reset_options) when the test ends, passed or failed."""
    _lib.check(_lib.lib().empose_set_option(name, value))


def gpu(x, dtype=torch.float32):
    return torch.as_tensor(np.asarray(x), dtype=dtype).to(DEV).contiguous()


@pytest.fixture(scope='module')
def big_model():
    return synthetic.make_model()


def build_net(case_or_cfg, model, vids=None, sd=None):
    smpl = SMPLLayer(model)
    net = create_model(case_or_cfg, smpl)
    if sd is not None:
        missing, unexpected = net.load_state_dict(H.sd_to_torch(sd), strict=False)
        assert not unexpected and all(k.startswith('smpl.') for k in missing)
    if vids is not None:
        net.vertex_ids = [int(v) for v in vids]
    return net.to(DEV).eval()


def cfg_of(meta, hidden=32):
    return lgd_config(int(meta['n_markers']), bool(meta['rnn']), int(meta['N']), hidden=hidden, rnn_hidden=hidden)


# ----------------------------------------------------------------------------------------------------------------------
def test_library_reports_gfx950():
    lib = _lib.lib()
    assert lib.empose_arch() == b'gfx950'
    assert torch.cuda.is_available()


@pytest.mark.parametrize('M,N,K', [(1, 4, 4), (64, 64, 32), (100, 66, 296), (257, 10, 512), (1000, 512, 144),
                                   (4096, 512, 512), (333, 200, 320), (130, 2048, 72),
                                   (12, 512, 2048), (1, 67, 1024), (16, 130, 1100),    # these three: gemm_fewrows_kernel
                                   (256, 512, 2048), (100, 64, 1024)])                 # eight-wave split-K
def test_linear_f32(M, N, K):
    rng = np.random.default_rng(M * 7 + N)
    lda = K + 8
    a = rng.normal(size=(M, lda)).astype(np.float32)
    w = rng.normal(size=(N, K)).astype(np.float32) / np.sqrt(K)
    scale = rng.uniform(0.5, 1.5, size=N).astype(np.float32)
    shift = rng.normal(size=N).astype(np.float32)
    # asymmetric operands: catches transposed outputs
    ref = (a[:, :K].astype(np.float64) @ w.astype(np.float64).T) * scale + shift
    ref = np.where(ref >= 0, ref, 0.25 * ref)
    A_, W_, out = gpu(a), gpu(w), torch.full((M, N + 3), -7.0, device=DEV)
    sc_, sh_ = gpu(scale), gpu(shift)  # keep every device tensor alive across the asynchronous call
    _lib.check(_lib.lib().empose_linear_f32(_lib.dptr(A_), lda, _lib.dptr(W_), K, _lib.dptr(out), N + 3, M, N, K,
                                            _lib.dptr(sc_), _lib.dptr(sh_), 1, 0.25, _lib.current_stream()))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    np.testing.assert_allclose(got[:, :N], ref, atol=2e-5 * np.sqrt(K / 32), rtol=1e-5)
    assert (got[:, N:] == -7.0).all()  # never writes outside N


def test_linear_f32_rejects_bad_arguments():
    x = torch.zeros(8, 8, device=DEV)
    rc = _lib.lib().empose_linear_f32(_lib.dptr(x), 8, _lib.dptr(x), 8, _lib.dptr(x), 8, 8, 8, 6, None, None, 0, 0.0,
                                      None)
    assert rc == -1 and b'multiples of 4' in _lib.lib().empose_last_error()


# ----------------------------------------------------------------------------------------------------------------------
def _smpl_case(model, vids, T, F, seed, n_markers):
    rng = np.random.default_rng(seed)
    theta = rng.normal(0, 0.25, size=(T, 66))
    beta = rng.normal(0, 1.0, size=(T, 10))
    W = T // F
    off_t = rng.normal(0, 0.02, size=(W, 12, 3))
    off_r = synthetic._exp_so3(rng.normal(0, 0.1, size=(W, 12, 3)))
    idx = list(range(12)) if n_markers == 12 else list(CONST.S_CONFIG_6)
    tab64 = TB.build_lgd_tables(model, vids, dtype=np.float64)
    rep = lambda a: np.repeat(a, F, axis=0)
    base = A.smpl_sensors(tab64, theta, beta, rep(off_r), rep(off_t))
    tgt_pos = base['pos'][:, idx] + rng.normal(0, 0.01, size=(T, len(idx), 3))
    tgt_ori = base['ori'][:, idx] @ synthetic._exp_so3(rng.normal(0, 0.05, size=(T, len(idx), 3)))
    scale = rng.choice([0.0, 1.0, 1.0, 1.0, 32.0 / 20.0], size=T)
    ref = A.smpl_sensors(tab64, theta, beta, rep(off_r), rep(off_t), tgt_pos, tgt_ori, idx, scale)
    tgt = np.concatenate([tgt_pos.reshape(T, -1), tgt_ori.reshape(T, -1)], axis=1)
    return theta, beta, off_r, off_t, tgt, scale, ref


@pytest.mark.parametrize('which,n_markers', [('small', 12), ('small', 6), ('big', 12), ('big', 6)])
def test_smpl_sensors_fwd_bwd(which, n_markers, big_model):
    if which == 'small':
        model, vids = H.small_model(), synthetic.small_vertex_ids(160)
    else:
        model, vids = big_model, CONST.VERTEX_IDS
    T, F = 96, 8
    theta, beta, off_r, off_t, tgt, scale, ref = _smpl_case(model, vids, T, F, 11, n_markers)
    net = build_net(lgd_config(n_markers, False, 1, hidden=32), model, vids)
    handle = net._ensure_handle(torch.device(DEV))
    lib = _lib.lib()
    th, be = gpu(theta), gpu(beta)
    pos, ori, joints = (torch.empty(T, n, device=DEV) for n in (36, 108, 66))
    g_th, g_be = torch.empty(T, 66, device=DEV), torch.empty(T, 10, device=DEV)
    nbytes = lib.empose_smpl_workspace_bytes(handle, T)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    tg, o_r, o_t, sc = gpu(tgt), gpu(off_r), gpu(off_t), gpu(scale)
    _lib.check(lib.empose_smpl_sensors_fwd_bwd(handle, T, F, _lib.dptr(th), 66, _lib.dptr(be), 10,
                                               _lib.dptr(o_r), _lib.dptr(o_t), _lib.dptr(tg),
                                               tg.shape[1], _lib.dptr(sc), _lib.dptr(pos), _lib.dptr(ori),
                                               _lib.dptr(joints), _lib.dptr(g_th), 66, _lib.dptr(g_be), 10,
                                               _lib.dptr(ws), nbytes, _lib.current_stream()))
    torch.cuda.synchronize()
    np.testing.assert_allclose(pos.cpu().numpy().reshape(T, 12, 3), ref['pos'], atol=1e-5)
    # frames are unit vectors built from differences of metre-scale fp32 positions over centimetre-scale edges: one ulp
    # of a position (1.2e-7) is 1e-5 of a frame entry
    np.testing.assert_allclose(ori.cpu().numpy().reshape(T, 12, 3, 3), ref['ori'], atol=5e-5)
    np.testing.assert_allclose(joints.cpu().numpy().reshape(T, 22, 3), ref['joints'], atol=1e-5)
    gmax = np.abs(ref['g_theta']).max()
    # fp32 against the float64 analytic oracle; measured worst case 3e-4 of the gradient scale (the residual direction
    # r / |r| amplifies the round-off of centimetre-scale differences of metre-scale positions)
    np.testing.assert_allclose(g_th.cpu().numpy(), ref['g_theta'], atol=5e-4 * max(gmax, 1.0), rtol=1e-3)
    np.testing.assert_allclose(g_be.cpu().numpy(), ref['g_beta'], atol=5e-4 * max(np.abs(ref['g_beta']).max(), 1.0),
                               rtol=1e-3)
    # frames with zero weight contribute an exactly-zero gradient
    assert (g_th.cpu().numpy()[scale == 0] == 0).all()


def test_smpl_sensors_large_batch_blend_gemm_path(big_model, monkeypatch):
    """At bench-size batches the blend-shape contraction runs on the row-block GEMM (csrc/mlp_fused.hip
    gemm_rows_kernel: A block resident in LDS, weights in fragment order).  Its first 96 frames must carry the bits of a
    96-frame call on the generic tile (same k order; the small call's own default, the split-K kernel, sums in another
    order and is compared at 1e-6), which is checked against the float64 blueprint above."""
    model, vids = big_model, CONST.VERTEX_IDS
    F, T_small, T_big = 8, 96, 128 * 200
    theta, beta, off_r, off_t, tgt, scale, ref = _smpl_case(model, vids, T_small, F, 11, 12)
    rng = np.random.default_rng(5)
    reps = T_big // T_small + 1
    tile = lambda a: np.concatenate([a] * reps, axis=0)
    theta_b = tile(theta)[:T_big] + np.concatenate([np.zeros((T_small, 66)), rng.normal(0, 0.05, size=(T_big - T_small, 66))])
    beta_b, tgt_b, scale_b = tile(beta)[:T_big], tile(tgt)[:T_big], tile(scale)[:T_big]
    off_r_b, off_t_b = tile(off_r)[:T_big // F], tile(off_t)[:T_big // F]
    net = build_net(lgd_config(12, False, 1, hidden=32), model, vids)
    handle = net._ensure_handle(torch.device(DEV))
    lib = _lib.lib()
    outs = {}
    small = (theta, beta, off_r, off_t, tgt, scale)
    _set_option(b'smpl_tile', 0)   # this test is about the general kernel behind both GEMM kernels
    for key, T, (th_, be_, or_, ot_, tg_, sc_) in (('splitk', T_small, small), (T_small, T_small, small),
                                                  (T_big, T_big, (theta_b, beta_b, off_r_b, off_t_b, tgt_b, scale_b))):
        _lib.check(lib.empose_set_option(b'gemm_splitk', 1 if key == 'splitk' else 0))   # kernel-variant switch
        th, be, o_r, o_t, tg, sc = gpu(th_), gpu(be_), gpu(or_), gpu(ot_), gpu(tg_), gpu(sc_)
        pos, ori, joints = (torch.empty(T, n, device=DEV) for n in (36, 108, 66))
        g_th, g_be = torch.empty(T, 66, device=DEV), torch.empty(T, 10, device=DEV)
        nbytes = lib.empose_smpl_workspace_bytes(handle, T)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
        _lib.check(lib.empose_smpl_sensors_fwd_bwd(handle, T, F, _lib.dptr(th), 66, _lib.dptr(be), 10, _lib.dptr(o_r),
                                                   _lib.dptr(o_t), _lib.dptr(tg), tg.shape[1], _lib.dptr(sc),
                                                   _lib.dptr(pos), _lib.dptr(ori), _lib.dptr(joints), _lib.dptr(g_th), 66,
                                                   _lib.dptr(g_be), 10, _lib.dptr(ws), nbytes, _lib.current_stream()))
        torch.cuda.synchronize()
        outs[key] = [t.cpu().numpy() for t in (pos, ori, joints, g_th, g_be)]
    np.testing.assert_allclose(outs[T_small][0].reshape(T_small, 12, 3), ref['pos'], atol=1e-5)
    for a, b, c in zip(outs[T_small], outs[T_big], outs['splitk']):
        assert np.array_equal(a, b[:T_small])     # same k order in both kernels: identical bits
        assert np.isfinite(b).all()
        # orientations and gradients amplify last-bit differences of the vertices (tolerances of the fwd/bwd test above)
        np.testing.assert_allclose(c, a, atol=2e-4 * max(1.0, float(np.abs(a).max())), rtol=1e-3)


def test_smpl_forward_only_matches(big_model):
    """tgt=NULL: positions/orientations/joints only (the final evaluation of the loop)."""
    T, F = 64, 32
    theta, beta, off_r, off_t, tgt, scale, ref = _smpl_case(big_model, CONST.VERTEX_IDS, T, F, 5, 12)
    net = build_net(lgd_config(12, False, 1, hidden=32), big_model)
    handle = net._ensure_handle(torch.device(DEV))
    lib = _lib.lib()
    pos, ori, joints = (torch.empty(T, n, device=DEV) for n in (36, 108, 66))
    nbytes = lib.empose_smpl_workspace_bytes(handle, T)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    th, be, o_r, o_t = gpu(theta), gpu(beta), gpu(off_r), gpu(off_t)
    _lib.check(lib.empose_smpl_sensors_fwd_bwd(handle, T, F, _lib.dptr(th), 66, _lib.dptr(be), 10,
                                               _lib.dptr(o_r), _lib.dptr(o_t), None, 0, None,
                                               _lib.dptr(pos), _lib.dptr(ori), _lib.dptr(joints), None, 0, None, 0,
                                               _lib.dptr(ws), nbytes, _lib.current_stream()))
    torch.cuda.synchronize()
    np.testing.assert_allclose(pos.cpu().numpy().reshape(T, 12, 3), ref['pos'], atol=1e-5)
    # rotation outputs are orthonormal frames times the offset rotation
    o = ori.cpu().numpy().reshape(T, 12, 3, 3).astype(np.float64)
    np.testing.assert_allclose(o @ np.swapaxes(o, -1, -2), np.broadcast_to(np.eye(3), o.shape), atol=1e-5)


# ----------------------------------------------------------------------------------------------------------------------
def test_update_nets_and_lstm_vs_oracle():
    case = H.load_case('lgdrnn12_n4_carry')
    model = H.small_model()
    net = build_net(cfg_of(case['meta']), model, case['meta']['vertex_ids'], case['sd'])
    handle = net._ensure_handle(torch.device(DEV))
    lib = _lib.lib()
    sd = H.sd_to_torch(case['sd'])
    rng = np.random.default_rng(0)
    T = 200
    x = rng.normal(size=(T, 296)).astype(np.float32)
    want_p = R.mlp_forward(sd, 'pose_net_iter.', torch.from_numpy(x)).numpy()
    want_s = R.mlp_forward(sd, 'shape_net_iter.', torch.from_numpy(x)).numpy()
    dp, ds = torch.empty(T, 66, device=DEV), torch.empty(T, 10, device=DEV)
    nbytes = lib.empose_update_workspace_bytes(handle, T)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    xg = gpu(x)
    _lib.check(lib.empose_update_nets_fwd(handle, T, _lib.dptr(xg), 296, _lib.dptr(dp), _lib.dptr(ds), _lib.dptr(ws),
                                          nbytes, _lib.current_stream()))
    torch.cuda.synchronize()
    np.testing.assert_allclose(dp.cpu().numpy(), want_p, atol=2e-5)
    np.testing.assert_allclose(ds.cpu().numpy(), want_s, atol=2e-5)

    # LSTM: ragged lengths + carried state
    B, F, Hh = 5, 13, 32
    xs = rng.normal(size=(B, F, 144)).astype(np.float32)
    lens = np.array([13, 7, 1, 13, 4], dtype=np.int32)
    h0 = rng.normal(size=(2, B, Hh)).astype(np.float32) * 0.3
    c0 = rng.normal(size=(2, B, Hh)).astype(np.float32) * 0.3
    want_y, (want_h, want_c) = R.lstm_forward(sd, 'rnn.lstm.', torch.from_numpy(xs), torch.from_numpy(lens).long(),
                                              (torch.from_numpy(h0), torch.from_numpy(c0)))
    y = torch.empty(B, F, Hh, device=DEV)
    hn, cn = torch.empty(2, B, Hh, device=DEV), torch.empty(2, B, Hh, device=DEV)
    nbytes = lib.empose_lstm_workspace_bytes(handle, B, F)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    xg, lg, h0g, c0g = gpu(xs), gpu(lens, torch.int32), gpu(h0), gpu(c0)
    _lib.check(lib.empose_lstm_fwd(handle, B, F, _lib.dptr(xg), 144, _lib.dptr(lg),
                                   _lib.dptr(h0g), _lib.dptr(c0g), _lib.dptr(y), _lib.dptr(hn), _lib.dptr(cn),
                                   _lib.dptr(ws), nbytes, _lib.current_stream()))
    torch.cuda.synchronize()
    np.testing.assert_allclose(y.cpu().numpy(), want_y.numpy(), atol=1e-5)
    np.testing.assert_allclose(hn.cpu().numpy(), want_h.numpy(), atol=1e-5)
    np.testing.assert_allclose(cn.cpu().numpy(), want_c.numpy(), atol=1e-5)


@pytest.mark.parametrize('hidden,skip,T', [(32, False, 12800 + 77), (512, False, 12800), (256, True, 13000),
                                           (100, False, 8192 + 5), (36, False, 9000)])
@pytest.mark.parametrize('x3', [1, 0], ids=['three_piece_bf16', 'fp32_mfma'])
def test_update_nets_large_batch_single_launch_path(hidden, skip, T, x3, monkeypatch):
    """Large batches run both update MLPs in ONE launch (csrc/mlp_fused.hip: a workgroup keeps 128 rows through all six
    layers): same results as the oracle's layer-by-layer MLP, incl. the ragged first layer (K = 296), the narrow output
    layers (66 / 10 columns), a ragged last row panel, skip connections, and the per-layer path on the first rows."""
    if x3 and (hidden % 64 != 0 or skip):
        pytest.skip('the three-piece kernel takes hidden widths of whole 64s without skip connections')
    _set_option(b'mlp_x3', x3)
    torch.manual_seed(hidden + T)
    cfg = lgd_config(12, False, 1, hidden=hidden, m_skip_connections=skip)
    net = create_model(cfg, SMPLLayer(H.small_model()))
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
    net.vertex_ids = synthetic.small_vertex_ids(160)
    net = net.to(DEV).eval()
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items() if not k.startswith('smpl.')}
    handle = net._ensure_handle(torch.device(DEV))
    lib = _lib.lib()
    x = torch.randn(T, 296, generator=g)
    with torch.no_grad():
        want_p = R.mlp_forward(sd, 'pose_net_iter.', x, skip=skip).numpy()
        want_s = R.mlp_forward(sd, 'shape_net_iter.', x, skip=skip).numpy()
    xg = x.to(DEV)
    outs = {}
    # 200 rows: too few row panels for the single launch -> the layer-by-layer kernels: on the generic tiles (same k order
    # as the fused kernel) and on their default, the split-K kernel
    for key, rows in ((T, T), (200, 200), ('splitk', 200)):
        _lib.check(lib.empose_set_option(b'gemm_splitk', 1 if key == 'splitk' else 0))   # kernel-variant switch
        dp, ds = torch.full((rows, 66), 7.0, device=DEV), torch.full((rows, 10), 7.0, device=DEV)
        nbytes = lib.empose_update_workspace_bytes(handle, rows)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
        _lib.check(lib.empose_update_nets_fwd(handle, rows, _lib.dptr(xg), 296, _lib.dptr(dp), _lib.dptr(ds),
                                              _lib.dptr(ws), nbytes, _lib.current_stream()))
        torch.cuda.synchronize()
        np.testing.assert_allclose(dp.cpu().numpy(), want_p[:rows], atol=ATOL)
        np.testing.assert_allclose(ds.cpu().numpy(), want_s[:rows], atol=ATOL)
        outs[key] = (dp.cpu().numpy(), ds.cpu().numpy())
    # both paths accumulate every dot product in the same k order: identical bits, not just close -- for the fp32-MFMA fused
    # kernel; the three-piece bf16 kernel (mlp_fused_x3.hip, what hidden widths of whole 64s take by default) forms the
    # same products in another order
    if lib.empose_get_option(b'mlp_x3') == 0 or hidden % 64 != 0 or skip:
        assert np.array_equal(outs[T][0][:200], outs[200][0]) and np.array_equal(outs[T][1][:200], outs[200][1])
    else:
        np.testing.assert_allclose(outs[T][0][:200], outs[200][0], atol=2e-5)
        np.testing.assert_allclose(outs[T][1][:200], outs[200][1], atol=2e-5)


def test_mlp_module_forward_vs_oracle():
    case = H.load_case('lgd12_n4')
    net = build_net(cfg_of(case['meta']), H.small_model(), case['meta']['vertex_ids'], case['sd'])
    x = torch.from_numpy(np.random.default_rng(3).normal(size=(7, 11, 144)).astype(np.float32))
    want = R.mlp_forward(H.sd_to_torch(case['sd']), 'pose_net_init.', x.reshape(-1, 144)).reshape(7, 11, 66)
    got = net.pose_net_init(x.to(DEV))
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), atol=2e-5)


# ----------------------------------------------------------------------------------------------------------------------
def _check_against_record(net, rec, res, B, F, N):
    pose = res['pose'].cpu().numpy()
    np.testing.assert_allclose(pose[:, :, 3:], rec['out_pose_hat'], atol=ATOL)
    np.testing.assert_allclose(pose[:, :, :3], rec['out_root_ori_hat'], atol=ATOL)
    np.testing.assert_allclose(res['shape'].cpu().numpy(), rec['out_shape_hat'], atol=ATOL)
    np.testing.assert_allclose(res['joints'].cpu().numpy(), rec['out_joints_hat'], atol=ATOL)
    for key, name in (('pose', 'pose'), ('shape', 'shape'), ('joints', 'joints'), ('markers', 'markers'),
                      ('markers_ori', 'markers_ori')):
        got = res['hist'][key].cpu().numpy().reshape(N + 1, B, F, -1)
        np.testing.assert_allclose(got, rec['hist_' + name], atol=ATOL)
    gp = res['trace']['g_pose'].cpu().numpy()
    gs = res['trace']['g_shape'].cpu().numpy()
    # values hooked out of the reference's autograd, O(10): 1e-5 of the tensor's scale (measured: at most 2.1e-6 of it,
    # 5.8e-5 absolute on values up to 27.5)
    np.testing.assert_allclose(gp, rec['g_pose'], atol=1e-5 * np.abs(rec['g_pose']).max(), rtol=0)
    np.testing.assert_allclose(gs, rec['g_shape'], atol=1e-5 * np.abs(rec['g_shape']).max(), rtol=0)


def _run_case(name, tags, sl_key=None):
    case = H.load_case(name)
    meta, w = case['meta'], case['in']
    N = int(meta['N'])
    net = build_net(cfg_of(meta), H.small_model(), meta['vertex_ids'], case['sd'])
    state = None
    for tag, (sf, ef) in tags:
        rec = case[tag]
        inp = H.oracle_inputs(w, sl=w[sl_key] if sl_key else None, sf=sf, ef=ef)
        B, F = inp['marker_pos'].shape[:2]
        res = net.forward_tensors(inp['marker_pos'].to(DEV), inp['marker_oris'].to(DEV), inp['offset_t'].to(DEV),
                                  inp['offset_r'].to(DEV),
                                  None if inp['marker_masks'] is None else inp['marker_masks'].to(DEV),
                                  inp['seq_lengths'].to(DEV), state=state, keep_history=True,
                                  keep_gradient_trace=True)
        torch.cuda.synchronize()
        _check_against_record(net, rec, res, B, F, N)
        state = res['state']
        if state is not None:
            np.testing.assert_allclose(state[0].cpu().numpy(), rec['rnn_h'], atol=2e-5)
            np.testing.assert_allclose(state[1].cpu().numpy(), rec['rnn_c'], atol=2e-5)


def test_golden_lgd12_no_rnn():
    _run_case('lgd12_n4', [('run', (None, None))])


def test_golden_lgdrnn12_two_chunks_with_state_carry():
    _run_case('lgdrnn12_n4_carry', [('chunk0', (0, 32)), ('chunk1', (32, 64))])


def test_golden_lgdrnn6():
    _run_case('lgdrnn6_n2', [('run', (None, None))])


def test_golden_ragged_and_masked():
    _run_case('lgdrnn12_n3_ragged_masked', [('run', (None, None))], sl_key='seq_lengths')


def test_golden_inner_windows_of_one_sequence():
    """SURVEY 8 a3, `forward(batch, window_size=k)` (reference models.py:146-163, 501): one 72-frame sequence fed to the
    reference as `net(batch, window_size=16)` over frames 0..40 (inner windows 16 / 16 / ragged 8, LSTM state handed from
    one inner window to the next, shape mean per inner window) and `net(batch, window_size=12, is_new_sequence=False)`
    over frames 40..72, with missing sensors.  The same two calls through this module's `forward` on the GPU: outputs,
    the merged N+1 histories, the gradient features of every inner window and the carried LSTM state."""
    from em_pose_amd.data.data import RealBatch
    case = H.load_case('lgdrnn12_n4_inner_windows')
    meta, w = case['meta'], case['in']
    N = int(meta['N'])
    net = build_net(cfg_of(meta), H.small_model(), meta['vertex_ids'], case['sd'])
    net.keep_gradient_trace = True
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    for tag, (sf, ef), new in (('call0', (0, 40), True), ('call1', (40, 72), False)):
        rec = case[tag]
        n = ef - sf
        b = RealBatch([0], torch.tensor([n]), t(w['poses'][:, sf:ef]), t(w['shapes']), torch.zeros(1, n, 3),
                      t(w['marker_pos'][:, sf:ef]), t(w['marker_oris'][:, sf:ef]), t(w['marker_masks'][:, sf:ef]),
                      t(w['offset_t']), t(w['offset_r'])).to_gpu(torch.device(DEV))
        out = net(b, window_size=int(rec['window_size']), is_new_sequence=new)
        torch.cuda.synchronize()
        for k in ('pose_hat', 'root_ori_hat', 'shape_hat', 'joints_hat'):
            assert out[k].shape == rec['out_' + k].shape
            np.testing.assert_allclose(out[k].cpu().numpy(), rec['out_' + k], atol=ATOL, err_msg=tag + k)
        for name in ('pose', 'shape', 'joints', 'markers', 'markers_ori'):
            hist = getattr(net, name + '_hat_history')
            assert len(hist) == N + 1
            got = np.stack([h.cpu().numpy().reshape(1, n, -1) for h in hist])
            np.testing.assert_allclose(got, rec['hist_' + name], atol=ATOL, err_msg=tag + name)
        assert net.markers_hat_history[0].shape == (1, n * 12, 3)
        assert len(net.gradient_trace) == 3          # one trace per inner window
        gp = np.concatenate([tr['g_pose'].cpu().numpy().reshape(N, -1, 66) for tr in net.gradient_trace], axis=1)
        gs = np.concatenate([tr['g_shape'].cpu().numpy().reshape(N, -1, 10) for tr in net.gradient_trace], axis=1)
        np.testing.assert_allclose(gp, rec['g_pose'], atol=1e-5 * np.abs(rec['g_pose']).max(), rtol=0)
        np.testing.assert_allclose(gs, rec['g_shape'], atol=1e-5 * np.abs(rec['g_shape']).max(), rtol=0)
        np.testing.assert_allclose(net.rnn.final_state[0].cpu().numpy(), rec['rnn_h'], atol=2e-5)
        np.testing.assert_allclose(net.rnn.final_state[1].cpu().numpy(), rec['rnn_c'], atol=2e-5)
    # the reference's restriction stands: a fresh one-entry `seq_lengths` per inner window only fits a batch of one
    b2 = RealBatch([0, 1], torch.tensor([40, 40]), t(np.repeat(w['poses'][:, :40], 2, 0)), t(np.repeat(w['shapes'], 2, 0)),
                   torch.zeros(2, 40, 3), t(np.repeat(w['marker_pos'][:, :40], 2, 0)),
                   t(np.repeat(w['marker_oris'][:, :40], 2, 0)), torch.ones(2, 40, 12),
                   t(np.repeat(w['offset_t'], 2, 0)), t(np.repeat(w['offset_r'], 2, 0))).to_gpu(torch.device(DEV))
    with pytest.raises((AssertionError, ValueError, _lib.EmposeError)):
        net(b2, window_size=16)


def test_module_forward_mirrors_reference_interface():
    """forward(batch) through the batch container: dict keys/shapes, history shapes, state carry (models.py:485-632)."""
    from em_pose_amd.data.data import RealBatch
    case = H.load_case('lgdrnn12_n4_carry')
    meta, w = case['meta'], case['in']
    net = build_net(cfg_of(meta), H.small_model(), meta['vertex_ids'], case['sd'])
    for tag, (sf, ef), new in (('chunk0', (0, 32), True), ('chunk1', (32, 64), False)):
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        b = RealBatch([0, 1], torch.tensor([32, 32]), t(w['poses'][:, sf:ef]), t(w['shapes']), torch.zeros(2, 32, 3),
                      t(w['marker_pos'][:, sf:ef]), t(w['marker_oris'][:, sf:ef]), torch.ones(2, 32, 12),
                      t(w['offset_t']), t(w['offset_r'])).to_gpu(torch.device(DEV))
        b.joints_gt = torch.zeros(2, 32, 66, device=DEV)  # set by the SMPLFK transform in the reference's flow
        out = net(b, is_new_sequence=new)
        rec = case[tag]
        for k in ('pose_hat', 'root_ori_hat', 'shape_hat', 'joints_hat'):
            assert out[k].shape == rec['out_' + k].shape
            np.testing.assert_allclose(out[k].cpu().numpy(), rec['out_' + k], atol=ATOL)
        assert len(net.pose_hat_history) == 5
        assert net.markers_hat_history[0].shape == (2, 32 * 12, 3)
        assert net.markers_ori_hat_history[0].shape == (2, 32 * 12 * 3, 3)
        assert net.joints_hat_history[0].shape == (2, 32 * 22, 3)
        _, loss_vals = net.backward(b, out)
        assert set(loss_vals) == {'pose', 'shape', 'reconstruction', 'fk', 'total_loss'}


def test_cpu_tensors_are_refused():
    case = H.load_case('lgd12_n4')
    net = build_net(cfg_of(case['meta']), H.small_model(), case['meta']['vertex_ids'], case['sd'])
    w = case['in']
    with pytest.raises(_lib.EmposeError):
        net.forward_tensors(torch.from_numpy(w['marker_pos']), torch.from_numpy(w['marker_oris']),
                            torch.from_numpy(w['offset_t']), torch.from_numpy(w['offset_r']))


# ----------------------------------------------------------------------------------------------------------------------
def _full_size_net(big_model, n_markers=12, rnn=True, N=4, seed=1615200973):
    torch.manual_seed(seed)
    net = create_model(lgd_config(n_markers, rnn, N), SMPLLayer(big_model))
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
    return net.eval()


def _oracle_sensors(big_model):
    bm = R.BodyModelTensors(big_model)
    tables = R.sensor_tables(big_model['f'], CONST.VERTEX_IDS)

    def fn(poses, betas, o_r, o_t):
        with torch.no_grad():
            p, o, _ = R.estimated_markers(bm, tables, CONST.VERTEX_IDS, torch.from_numpy(poses),
                                          torch.from_numpy(betas), torch.from_numpy(o_r), torch.from_numpy(o_t))
        return p.numpy(), o.numpy()
    return bm, tables, fn


@pytest.mark.parametrize('n_markers,rnn,N', [(12, True, 4), (12, False, 4), (6, True, 2)])
def test_full_size_model_vs_oracle(big_model, n_markers, rnn, N):
    """The released architectures (2x512 nets, 2x512 LSTM, V=6890 body model, real sensor vertex ids)."""
    net = _full_size_net(big_model, n_markers, rnn, N)
    bm, tables, fn = _oracle_sensors(big_model)
    B, F = 3, 32
    w = synthetic.make_windows(B, F, 1234 + n_markers, fn)
    inp = H.oracle_inputs(w)
    sd = {k: v for k, v in net.state_dict().items() if not k.startswith('smpl.')}
    want, hist = R.ief_forward(sd, bm, tables, CONST.VERTEX_IDS, inp, n_markers=n_markers, N=N, rnn_init=rnn)
    net = net.to(DEV)
    res = net.forward_tensors(*(inp[k].to(DEV) for k in ('marker_pos', 'marker_oris', 'offset_t', 'offset_r')),
                              keep_history=True, keep_gradient_trace=True)
    torch.cuda.synchronize()
    pose = res['pose'].cpu().numpy()
    np.testing.assert_allclose(pose[:, :, 3:], want['pose_hat'].numpy(), atol=ATOL)
    np.testing.assert_allclose(pose[:, :, :3], want['root_ori_hat'].numpy(), atol=ATOL)
    np.testing.assert_allclose(res['shape'].cpu().numpy(), want['shape_hat'].numpy(), atol=ATOL)
    np.testing.assert_allclose(res['joints'].cpu().numpy(), want['joints_hat'].numpy(), atol=ATOL)
    got_m = res['hist']['markers'].cpu().numpy().reshape(N + 1, B * F, 12, 3)
    np.testing.assert_allclose(got_m, np.stack([t.numpy() for t in hist['markers']]), atol=ATOL)
    # MPJPE between the two implementations, in millimetres (north_star: within 0.1 mm)
    dj = (res['joints'].cpu().numpy() - want['joints_hat'].numpy()).reshape(B, F, 22, 3)
    assert np.linalg.norm(dj, axis=-1).mean() * 1000.0 < 0.1


def test_full_size_properties_shard_invariance_and_determinism(big_model):
    """BASELINE config 3 size (B=1024, ws=32): results for a window do not depend on its position in the batch or on
    which other windows share the launch (the multi-GPU sharding of SURVEY.md 8e relies on exactly this)."""
    net = _full_size_net(big_model).to(DEV)
    _, _, fn = _oracle_sensors(big_model)
    w = synthetic.make_windows(16, 32, 99, fn)
    rep = 64
    big = {k: np.concatenate([w[k]] * rep, axis=0) for k in ('marker_pos', 'marker_oris', 'offset_t', 'offset_r')}
    args = lambda d, s: [torch.from_numpy(d[k][s]).to(DEV) for k in ('marker_pos', 'marker_oris', 'offset_t',
                                                                      'offset_r')]
    full = net.forward_tensors(*args(big, slice(None)))
    torch.cuda.synchronize()
    assert full['pose'].shape == (1024, 32, 66)
    assert torch.isfinite(full['pose']).all() and torch.isfinite(full['joints']).all()
    small = net.forward_tensors(*args(w, slice(None)))
    for k in ('pose', 'shape', 'joints'):
        a = full[k].cpu().numpy()
        # every replica of the 16 windows gives the same answer, and equals the 16-window launch
        np.testing.assert_allclose(a.reshape(rep, 16, 32, -1), np.broadcast_to(a[:16], (rep, 16, 32, a.shape[-1])),
                                   atol=1e-6)
        np.testing.assert_allclose(a[:16], small[k].cpu().numpy(), atol=1e-6)
    # shape is constant inside a window (m_average_shape)
    s = full['shape'].cpu().numpy()
    np.testing.assert_allclose(s, np.broadcast_to(s[:, :1], s.shape), atol=1e-6)


# ----------------------------------------------------------------------------------------------------------------------
def test_full_mesh_vertices_vs_oracle(big_model):
    smpl = SMPLLayer(big_model).to(DEV)
    rng = np.random.default_rng(2)
    n = 70
    pose = rng.normal(0, 0.3, size=(n, 63)).astype(np.float32)
    root = rng.normal(0, 0.5, size=(n, 3)).astype(np.float32)
    betas = rng.normal(0, 1, size=(n, 16)).astype(np.float32)
    trans = rng.normal(0, 1, size=(n, 3)).astype(np.float32)
    bm = R.BodyModelTensors(big_model)
    v_ref, j_ref = R.smpl_fk(bm, torch.from_numpy(pose), torch.from_numpy(betas), torch.from_numpy(root),
                             torch.from_numpy(trans))
    v, j = smpl(poses_body=gpu(pose), betas=gpu(betas), poses_root=gpu(root), trans=gpu(trans))
    np.testing.assert_allclose(v.cpu().numpy(), v_ref.numpy(), atol=2e-5)
    assert tuple(j.shape) == (n, 52, 3)   # all 52 posed joints, as the reference's `body.Jtr` (smpl.py:121-122)
    np.testing.assert_allclose(j.cpu().numpy(), j_ref.numpy(), atol=2e-5)
    # zero pose / zero shape reproduces the template; broadcast of a single beta row
    v0, j0 = smpl(poses_body=torch.zeros(2, 63, device=DEV), betas=torch.zeros(10, device=DEV))
    np.testing.assert_allclose(v0[0].cpu().numpy(), big_model['v_template'], atol=1e-6)
    np.testing.assert_allclose(v0[1].cpu().numpy(), v0[0].cpu().numpy(), atol=0)


def test_full_mesh_ragged_sizes_and_many_bones():
    """mesh_rows_kernel edge cases: 1 frame, frame counts that are not multiples of the 64-frame block (one and several
    blocks per CU column), no translation, and a body model whose vertices are skinned by up to
    six bones (the kernel keeps four in registers and loops over the rest)."""
    model = dict(H.small_model())
    V = model['v_template'].shape[0]
    rng = np.random.default_rng(11)
    w = np.array(model['weights'], dtype=np.float64, copy=True)
    for v in range(0, V, 3):                      # every third vertex: six bones
        bones = rng.choice(22, size=6, replace=False)
        w[v] = 0
        w[v, bones] = rng.uniform(0.1, 1.0, size=6)
        w[v] /= w[v].sum()
    model['weights'] = w.astype(model['weights'].dtype)
    from em_pose_amd.bodymodels import tables as TB
    assert TB.build_full_mesh_tables(model)['kb'] == 6
    bm = R.BodyModelTensors(model)
    smpl = SMPLLayer(model).to(DEV)
    for n, with_trans in ((1, True), (131, False), (700, True)):
        pose = rng.normal(0, 0.4, size=(n, 63)).astype(np.float32)
        root = rng.normal(0, 0.5, size=(n, 3)).astype(np.float32)
        betas = rng.normal(0, 1, size=(n, 10)).astype(np.float32)
        trans = rng.normal(0, 1, size=(n, 3)).astype(np.float32) if with_trans else None
        v_ref, j_ref = R.smpl_fk(bm, torch.from_numpy(pose), torch.from_numpy(betas), torch.from_numpy(root),
                                 torch.from_numpy(trans) if with_trans else None)
        v, j = smpl(poses_body=gpu(pose), betas=gpu(betas), poses_root=gpu(root),
                    trans=gpu(trans) if with_trans else None)
        np.testing.assert_allclose(v.cpu().numpy(), v_ref.numpy(), atol=2e-5)
        np.testing.assert_allclose(j.cpu().numpy(), j_ref.numpy(), atol=2e-5)


def test_full_mesh_split_bf16_variant(big_model):
    """The explicitly selected split-bf16 variant of the blend-shape contraction (mesh_rows_bf16_kernel,
    `SMPLLayer(arithmetic='bf16x3')`, `bench.py --workload vertices --arith bf16x3`): vertices within 1e-4 m of the
    float64-free oracle (the bar VERDICT r1 item 7 sets; measured error is two orders below), joints bit-identical to
    the fp32 path (the kinematic chain does not change), same edge cases as the fp32 kernel."""
    rng = np.random.default_rng(21)
    bm = R.BodyModelTensors(big_model)
    fast = SMPLLayer(big_model, arithmetic='bf16x3').to(DEV)
    exact = SMPLLayer(big_model).to(DEV)
    for n, with_trans in ((1, True), (70, True), (131, False)):
        pose = rng.normal(0, 0.5, size=(n, 63)).astype(np.float32)
        root = rng.normal(0, 0.5, size=(n, 3)).astype(np.float32)
        betas = rng.normal(0, 1.5, size=(n, 16)).astype(np.float32)
        trans = rng.normal(0, 1, size=(n, 3)).astype(np.float32) if with_trans else None
        v_ref, j_ref = R.smpl_fk(bm, torch.from_numpy(pose), torch.from_numpy(betas), torch.from_numpy(root),
                                 torch.from_numpy(trans) if with_trans else None)
        kw = dict(poses_body=gpu(pose), betas=gpu(betas), poses_root=gpu(root), trans=gpu(trans) if with_trans else None)
        v, j = fast(**kw)
        v32, j32 = exact(**kw)
        assert np.abs(v.cpu().numpy() - v_ref.numpy()).max() < 1e-4
        np.testing.assert_allclose(v.cpu().numpy(), v32.cpu().numpy(), atol=2e-5)
        assert torch.equal(j, j32)
    v0, _ = fast(poses_body=torch.zeros(2, 63, device=DEV), betas=torch.zeros(10, device=DEV))
    np.testing.assert_allclose(v0[0].cpu().numpy(), big_model['v_template'], atol=1e-6)   # three pieces: fp32-exact
    # repeated launches are bit-identical (a two-waves-per-SIMD build of this kernel was not: mesh.hip), both kernels
    n = 4096
    g = torch.Generator().manual_seed(5)
    kw = dict(poses_body=(torch.randn(n, 63, generator=g) * 0.5).to(DEV), betas=torch.randn(n, 10, generator=g).to(DEV),
              poses_root=(torch.randn(n, 3, generator=g) * 0.5).to(DEV))
    for layer in (fast, exact):
        first = layer(**kw)[0].clone()
        for _ in range(5):
            assert torch.equal(layer(**kw)[0], first)
    assert float((fast(**kw)[0] - exact(**kw)[0]).abs().max()) < 2e-5
    # a body model with six bones per vertex (the EXTRA instantiation), frame counts off the 64-frame block
    model = dict(H.small_model())
    V = model['v_template'].shape[0]
    w = np.array(model['weights'], dtype=np.float64, copy=True)
    for vtx in range(0, V, 3):
        bones = rng.choice(22, size=6, replace=False)
        w[vtx] = 0
        w[vtx, bones] = rng.uniform(0.1, 1.0, size=6)
        w[vtx] /= w[vtx].sum()
    model['weights'] = w.astype(model['weights'].dtype)
    bm = R.BodyModelTensors(model)
    fast = SMPLLayer(model, arithmetic='bf16x3').to(DEV)
    n = 700
    pose = rng.normal(0, 0.4, size=(n, 63)).astype(np.float32)
    root = rng.normal(0, 0.5, size=(n, 3)).astype(np.float32)
    betas = rng.normal(0, 1, size=(n, 10)).astype(np.float32)
    v_ref, _ = R.smpl_fk(bm, torch.from_numpy(pose), torch.from_numpy(betas), torch.from_numpy(root), None)
    v, _ = fast(poses_body=gpu(pose), betas=gpu(betas), poses_root=gpu(root))
    assert np.abs(v.cpu().numpy() - v_ref.numpy()).max() < 1e-4
    with pytest.raises(ValueError):
        SMPLLayer(model, arithmetic='bf16')


# ----------------------------------------------------------------------------------------------------------------------
def test_streaming_evaluation_driver_matches_oracle_chunk_by_chunk():
    """evaluate_real's loop: one 600-frame recording, 256-frame chunks, LSTM state carried chunk to chunk, missing
    sensors; the model outputs equal the oracle run chunk by chunk, and the metrics equal a NumPy recomputation."""
    from em_pose_amd.data.data import RealBatch, RealSample
    from em_pose_amd.data.transforms import NormalizeRealMarkers, NormalizeRoot, ToTensor
    from em_pose_amd.eval.helpers import evaluate_sequences, window_generator
    from em_pose_amd.eval.metrics import MetricsEngine
    case = H.load_case('lgdrnn6_n2')
    meta = case['meta']
    model = H.small_model()
    vids = [int(v) for v in meta['vertex_ids']]
    net = build_net(cfg_of(meta), model, vids, case['sd'])
    net.keep_history = False
    smpl = SMPLLayer(model).to(DEV)
    bm = R.BodyModelTensors(model)
    tables = R.sensor_tables(model['f'], vids)

    def sensors(poses, betas, o_r, o_t):
        with torch.no_grad():
            p, o, _ = R.estimated_markers(bm, tables, vids, torch.from_numpy(poses), torch.from_numpy(betas),
                                          torch.from_numpy(o_r), torch.from_numpy(o_t))
        return p.numpy(), o.numpy()
    d = synthetic.make_sequence(600, 5, sensors, missing_rate=0.01)
    s = RealSample('rec', d['sensor_pos'], d['sensor_oris'], d['sensor_masks'].astype(np.float32), d['smpl_poses'],
                   d['smpl_shape'], d['smpl_trans'], {'means': d['offset_means'], 'covs': d['offset_covs'],
                                                      'r': d['offset_r']})
    batch = NormalizeRoot()(RealBatch.from_sample_list([ToTensor()(NormalizeRealMarkers()(s))]))

    # oracle, chunk by chunk with state carry
    sd = H.sd_to_torch(case['sd'])
    state, want = None, []
    for chunk in window_generator(batch, 256):
        inp = chunk.get_inputs()
        inp['seq_lengths'] = chunk.seq_lengths.long()
        out, hist = R.ief_forward(sd, bm, tables, vids, inp, n_markers=6, N=int(meta['N']), rnn_init=True,
                                  rnn_state=state)
        state = hist['rnn_state']
        want.append(out)
    me, per_seq, frames = evaluate_sequences(net, [batch], smpl, torch.device(DEV), window_size=256)
    assert frames == 600 and len(per_seq) == 1
    # re-run the model chunks to compare raw outputs
    for c, chunk in enumerate(window_generator(batch, 256)):
        out = net(chunk.to_gpu(torch.device(DEV)), is_new_sequence=(c == 0))
        for k in ('pose_hat', 'root_ori_hat', 'shape_hat', 'joints_hat'):
            np.testing.assert_allclose(out[k].cpu().numpy(), want[c][k].numpy(), atol=ATOL)
    m = me.get_metrics()
    assert all(np.isfinite(v) for v in m.values()) and m['MPJPE [mm]'] > 0
    # the frames with a missing sensor are excluded from the metrics (frame_mask semantics of metrics.py:166-181)
    n_valid = int((d['sensor_masks'].all(axis=1)).sum())
    assert np.concatenate(me.eucl_dists).shape[0] == n_valid
    # MPJPE recomputed with the oracle's FK on the same predictions
    pose = torch.cat([w['pose_hat'] for w in want], 1)[0]
    root = torch.cat([w['root_ori_hat'] for w in want], 1)[0]
    shape0 = want[0]['shape_hat'][0, :1].expand(600, 10)
    valid = torch.from_numpy(d['sensor_masks'].all(axis=1))
    _, j_hat = R.smpl_fk(bm, pose, shape0, root)
    _, j_gt = R.smpl_fk(bm, batch.poses[0, :, 3:], batch.shapes.expand(600, 10), batch.poses[0, :, :3])
    e = np.linalg.norm((j_hat[:, :22] - j_gt[:, :22]).numpy()[valid.numpy()], axis=-1)
    want_mpjpe = float(np.mean(np.mean(e, axis=0)[me.eucl_idxs]) * 1000.0)
    assert m['MPJPE [mm]'] == pytest.approx(want_mpjpe, rel=1e-3)


def test_virtual_marker_helper_vs_reference_vectors():
    """get_virtual_pos_and_rot on arbitrary vertices: against vectors recorded from the reference's own helper."""
    import os
    from em_pose_amd.data.virtual_sensors import VirtualMarkerHelper
    z = np.load(os.path.join(H.GOLDEN, 'components.npz'))
    vids = synthetic.small_vertex_ids(160)
    helper = VirtualMarkerHelper(SMPLLayer(H.small_model()))
    pos, ori, nor = helper.get_virtual_pos_and_rot(gpu(z['vs_verts']), vids)
    np.testing.assert_allclose(pos.cpu().numpy(), z['vs_pos'], atol=1e-7)
    np.testing.assert_allclose(ori.cpu().numpy(), z['vs_ori'], atol=5e-6)
    np.testing.assert_allclose(nor.cpu().numpy(), z['vs_nor'], atol=1e-8)
    assert helper.get_vertex_helpers(vids) == z['vs_helpers'].tolist()


# ----------------------------------------------------------------------------------------------------------------------
# training path (BASELINE configs[4])
# ----------------------------------------------------------------------------------------------------------------------
def test_smpl_sensors_vjp_vs_autograd(big_model):
    """Vector-Jacobian product of (pos, ori, joints) w.r.t. (pose, shape) for arbitrary cotangents."""
    T, F = 40, 8
    rng = np.random.default_rng(21)
    theta = rng.normal(0, 0.25, size=(T, 66)).astype(np.float32)
    beta = rng.normal(0, 1.0, size=(T, 10)).astype(np.float32)
    off_t = rng.normal(0, 0.02, size=(T // F, 12, 3)).astype(np.float32)
    off_r = synthetic._exp_so3(rng.normal(0, 0.1, size=(T // F, 12, 3))).astype(np.float32)
    d_pos = rng.normal(size=(T, 12, 3)).astype(np.float32)
    d_ori = rng.normal(size=(T, 12, 3, 3)).astype(np.float32)
    d_j = rng.normal(size=(T, 22, 3)).astype(np.float32)
    bm = R.BodyModelTensors(big_model, dtype=torch.float64)
    tables = R.sensor_tables(big_model['f'], CONST.VERTEX_IDS)
    th = torch.from_numpy(theta).double().requires_grad_(True)
    be = torch.from_numpy(beta).double().requires_grad_(True)
    rep = lambda a: torch.from_numpy(np.repeat(a, F, axis=0)).double()
    pos, ori, joints = R.estimated_markers(bm, tables, CONST.VERTEX_IDS, th, be, rep(off_r), rep(off_t))
    obj = (pos * torch.from_numpy(d_pos)).sum() + (ori * torch.from_numpy(d_ori)).sum() + \
        (joints * torch.from_numpy(d_j)).sum()
    want_th, want_be = torch.autograd.grad(obj, [th, be])

    from em_pose_amd.nn.models import _SmplSensorsFn
    net = build_net(lgd_config(12, False, 1, hidden=32), big_model)
    net.train()
    p = gpu(theta).requires_grad_(True)
    s = gpu(beta).requires_grad_(True)
    o_r, o_t = gpu(off_r), gpu(off_t)
    got_pos, got_ori, got_j = _SmplSensorsFn.apply(net, p, s, o_r, o_t, F)
    np.testing.assert_allclose(got_pos.detach().cpu().numpy(), pos.detach().numpy(), atol=1e-5)
    (got_pos * gpu(d_pos)).sum().add((got_ori * gpu(d_ori)).sum()).add((got_j * gpu(d_j)).sum()).backward()
    np.testing.assert_allclose(p.grad.cpu().numpy(), want_th.numpy(), atol=2e-4 * want_th.abs().max().item(), rtol=1e-3)
    np.testing.assert_allclose(s.grad.cpu().numpy(), want_be.numpy(), atol=2e-4 * want_be.abs().max().item(), rtol=1e-3)


@pytest.mark.parametrize('M,Cn', [(384, 512), (48, 32), (7, 100), (2, 5), (8192, 512), (3001, 100)])
def test_bn_prelu_train_function_vs_torch_autograd(M, Cn):
    """nn/layers.py::bn_prelu_train (train-mode BatchNorm1d + PReLU, one kernel forward, one backward) against the torch
    modules and their autograd in float64: output, dx, dgamma, dbeta, dslope, running statistics, batch counter."""
    from em_pose_amd.nn.layers import bn_prelu_train
    torch.manual_seed(M * 7 + Cn)
    bn, act = torch.nn.BatchNorm1d(Cn).to(DEV).train(), torch.nn.PReLU().to(DEV)
    with torch.no_grad():
        bn.weight.uniform_(0.2, 1.5); bn.bias.normal_(0, 0.3); act.weight.fill_(0.17)
        bn.running_mean.normal_(0, 0.1); bn.running_var.uniform_(0.5, 1.5)
    bn64, act64 = torch.nn.BatchNorm1d(Cn).double().to(DEV).train(), torch.nn.PReLU().double().to(DEV)
    bn64.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in bn.state_dict().items()})
    act64.load_state_dict({k: v.double() for k, v in act.state_dict().items()})
    x = (torch.randn(M, Cn, device=DEV) * 2 + 0.5).requires_grad_(True)
    dz = torch.randn(M, Cn, device=DEV)
    z = bn_prelu_train(x, bn, act)
    z.backward(dz)
    x64 = x.detach().double().requires_grad_(True)
    z64 = act64(bn64(x64))
    z64.backward(dz.double())
    tol = lambda w: 5e-5 * max(1.0, float(w.abs().max()))
    for g, w in ((z, z64), (x.grad, x64.grad), (bn.weight.grad, bn64.weight.grad), (bn.bias.grad, bn64.bias.grad),
                 (act.weight.grad, act64.weight.grad), (bn.running_mean, bn64.running_mean),
                 (bn.running_var, bn64.running_var)):
        np.testing.assert_allclose(g.detach().cpu().numpy(), w.detach().cpu().numpy(), atol=tol(w))
    assert int(bn.num_batches_tracked) == 1


@pytest.mark.parametrize('M,K,N', [(384, 512, 512), (384, 296, 512), (384, 512, 66), (5, 8, 3), (100, 20, 10)])
def test_linear_train_function_vs_torch_autograd(M, K, N):
    """nn/layers.py::linear_train (forward + dX + dW + db on the strided split-K GEMM) against torch.nn.functional.linear
    and its autograd; strided operands of every transposition case through the C entry point."""
    from em_pose_amd.nn.layers import linear_train
    torch.manual_seed(M + K + N)
    lin = torch.nn.Linear(K, N).to(DEV)
    x = torch.randn(M, K, device=DEV, requires_grad=True)
    dy = torch.randn(M, N, device=DEV)
    y = linear_train(x, lin)
    y.backward(dy)
    got = [y.detach(), x.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone()]
    x.grad = None
    lin.zero_grad()
    y2 = torch.nn.functional.linear(x.double(), lin.weight.double(), lin.bias.double())
    y2.backward(dy.double())
    want = [y2.detach(), x.grad, lin.weight.grad, lin.bias.grad]
    for g, w in zip(got, want):
        np.testing.assert_allclose(g.cpu().numpy(), w.cpu().double().numpy(), atol=3e-5 * max(1.0, float(w.abs().max())))
    lib = _lib.lib()
    assert lib.empose_gemm_strided_applicable(384, 512) == 1 and lib.empose_gemm_strided_applicable(4096, 512) == 0
    out = torch.empty(M, N, device=DEV)
    assert lib.empose_gemm_strided_f32(0, N, K, _lib.dptr(x), K, 1, _lib.dptr(lin.weight), K, 1, _lib.dptr(out), N, None,
                                       None) != 0


@pytest.mark.parametrize('fused', [False, True, 'epi', 'cols'],
                         ids=['bn_kernels', 'bn_in_gemms', 'bn_epilogue_finish', 'one_launch_layers'])
@pytest.mark.parametrize('name', ['train_lgdrnn12_n2', 'train_lgd6_n2'])
def test_training_step_matches_reference_gradients(name, fused):
    """forward (train mode) + backward: losses and EVERY parameter gradient against the reference's own training
    step recorded in tests/golden (incl. the in-forward E.backward() deposits, ragged lengths, train-mode BatchNorm).
    `fused`: the BatchNorm / PReLU passes folded into the GEMMs (csrc/train_fused.hip; off by default -- measured no
    faster --, forced here onto the recorded 48-frame batch)."""
    from em_pose_amd.data.data import SyntheticBatch
    _set_option(b'train_fused', 2 if fused is True else 0)
    _set_option(b'train_epi', 2 if fused == 'epi' else 0)   # round 4: statistics in the GEMM epilogues + one finish launch
    _set_option(b'train_cols', 1 if fused == 'cols' else 0)   # round 5: product + BatchNorm of both networks as one launch
    case = H.load_case(name)
    meta, w, rec = case['meta'], case['in'], case['run']
    net = build_net(cfg_of(meta), H.small_model(), meta['vertex_ids'], case['sd'])
    net.train()
    batch = SyntheticBatch(w, torch.from_numpy(w['seq_lengths']).to(DEV), device=DEV)
    batch.joints_gt = gpu(w['joints_gt'])
    net.zero_grad()
    out = net(batch)
    assert net._engine is not None   # the hand-written forward / reverse sweep (nn/train_engine.py), not autograd
    total, loss_vals = net.backward(batch, out)
    # Train mode is ill-conditioned (residual direction r/|r|, BatchNorm statistics over 48 frames): the reference's own
    # outputs move by `sens` when its inputs change by one unit in the last place (tests/golden/train_sensitivity.json,
    # recorded from the reference). The tolerance is the north-star 1e-4 or four such units, whichever is larger.
    import json
    with open(os.path.join(H.GOLDEN, 'train_sensitivity.json')) as f:
        sens = json.load(f)[name]
    for k in ('pose_hat', 'root_ori_hat', 'shape_hat', 'joints_hat'):
        np.testing.assert_allclose(out[k].detach().cpu().numpy(), rec['out_' + k], atol=max(ATOL, 4.0 * sens[k]))
    for k in ('pose', 'shape', 'reconstruction', 'fk', 'total_loss'):
        assert loss_vals[k] == pytest.approx(float(rec['loss_' + k]), rel=2e-4, abs=1e-6), k
    checked = 0
    # biases in front of a train-mode BatchNorm have a mathematically zero gradient (pure round-off in both
    # implementations), so the absolute tolerance is tied to the overall gradient scale, not to each tensor's own
    gmax = max(np.abs(v).max() for kk, v in rec.items() if kk.startswith('grad/'))
    for k, p in net.named_parameters():
        if k.startswith('smpl.'):
            continue
        want = rec.get('grad/' + k)
        if want is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        got = p.grad.detach().cpu().numpy()
        pre_bn_bias = k.endswith('.bias') and ('input_to_hidden' in k or '.layers.0.' in k or '.layers.4.' in k)
        if pre_bn_bias:  # mathematically zero in both implementations: only check that it is round-off
            assert np.abs(got).max() < 1e-4 * gmax and np.abs(want).max() < 1e-4 * gmax, k
            continue
        # 1e-4 of the tensor's scale (measured: at most 5.5e-5, scripts/dev/measure_train_grad_error.py), or four times
        # what the reference's own gradient moves under a one-ulp change of its inputs (train_sensitivity.json; measured:
        # at most 2.3 such units), whichever is larger.  No relative term.
        tol = max(1e-4 * max(np.abs(want).max(), 1e-4 * gmax), 4.0 * sens['grad'].get(k, 0.0))
        np.testing.assert_allclose(got, want, atol=tol, rtol=0, err_msg=k)
        checked += 1
    assert checked >= 14
    for k, v in net.state_dict().items():
        if 'running_' in k:  # BatchNorm running statistics were updated like the reference's
            np.testing.assert_allclose(v.cpu().numpy(), rec['after/' + k], atol=1e-5, err_msg=k)
    # an optimiser step invalidates nothing: the body-model handle is independent of the network weights
    h1 = net._smpl_handle.value
    torch.optim.Adam([p for n, p in net.named_parameters() if not n.startswith('smpl.')], lr=1e-3).step()
    net.zero_grad()
    net.backward(batch, net(batch))
    assert net._smpl_handle.value == h1


def test_training_weight_gradients_once_over_all_iterations_equal_per_iteration_sums():
    """The reverse sweep forms dW / db of the update networks once over the N applications (empose_mlp_train_wgrad: one
    A^T B product of N * T rows per layer, row segments addressed in the kernel); the per-application products + sums
    (empose_mlp_train_bwd with accumulate) give the same gradients up to the order of the additions."""
    from em_pose_amd.data.data import SyntheticBatch
    from em_pose_amd.nn.train_engine import LgdTrainEngine
    case = H.load_case('train_lgdrnn12_n2')
    meta = dict(case['meta'])
    net = build_net(cfg_of(meta), H.small_model(), meta['vertex_ids'], case['sd'])
    net.train()
    g = torch.Generator().manual_seed(9)
    Bn, Fn = 6, 32                      # 192 rows: on the 32-row grid, so the batched product is taken
    w = {k: v for k, v in case['in'].items()}
    rep = lambda a: np.concatenate([a] * 8, axis=0)[:Bn] if a.shape[0] < Bn else a[:Bn]
    def fit(a):   # tile the recorded window batch along batch and time to (6, 32, ...)
        a = rep(a)
        if a.ndim >= 2 and a.shape[1] == case['in']['marker_pos'].shape[1]:
            a = np.concatenate([a] * 4, axis=1)[:, :Fn]
        return np.ascontiguousarray(a)
    w = {k: fit(v) for k, v in w.items() if k != 'seq_lengths'}
    batch = SyntheticBatch(w, torch.tensor([32, 32, 20, 32, 7, 32], device=DEV), device=DEV)
    batch.joints_gt = gpu(w['joints_gt'])
    grads = {}
    bn_state = {k: v.clone() for k, v in net.state_dict().items() if 'running_' in k or 'num_batches' in k}
    for mode in (True, False):
        LgdTrainEngine.batched_wgrad = mode
        try:
            net.load_state_dict(bn_state, strict=False)
            net.zero_grad()
            net.backward(batch, net(batch))
            grads[mode] = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
        finally:
            LgdTrainEngine.batched_wgrad = True
    assert len(grads[True]) >= 14
    gmax = max(float(v.abs().max()) for v in grads[False].values())
    for k, v in grads[False].items():
        np.testing.assert_allclose(grads[True][k].cpu().numpy(), v.cpu().numpy(),
                                   atol=2e-5 * max(float(v.abs().max()), 1e-3 * gmax), rtol=1e-4, err_msg=k)


def test_training_weight_gradients_long_reduction_of_the_full_width_networks():
    """The released width (512 hidden units) at 256 windows: the one-product-per-layer weight gradients run the
    long-reduction variant of A^T B over row segments (4 x 8192 rows, 512 workgroups of 16-row chunks); the
    per-application products give the same gradients up to the order of the additions."""
    from em_pose_amd.data.data import SyntheticBatch
    from em_pose_amd.nn.train_engine import LgdTrainEngine
    model = H.small_model()
    vids = H.load_case('train_lgdrnn12_n2')['meta']['vertex_ids']
    B, F = 256, 32
    torch.manual_seed(3)
    net = build_net(lgd_config(12, False, 4), model, vids).train()
    w = synthetic.make_windows(B, F, 11)
    g = torch.Generator().manual_seed(11)
    w['marker_pos'] = torch.randn(B, F, 36, generator=g).numpy()
    w['marker_oris'] = torch.randn(B, F, 108, generator=g).numpy()
    batch = SyntheticBatch(w, device=DEV)
    batch.joints_gt = torch.randn(B, F, 66, generator=g).to(DEV)
    bn_state = {k: v.clone() for k, v in net.state_dict().items() if 'running_' in k or 'num_batches' in k}
    grads = {}
    for mode in (True, False):
        LgdTrainEngine.batched_wgrad = mode
        try:
            net.load_state_dict(bn_state, strict=False)
            net.zero_grad()
            net.backward(batch, net(batch))
            grads[mode] = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
        finally:
            LgdTrainEngine.batched_wgrad = True
    gmax = max(float(v.abs().max()) for v in grads[False].values())
    assert gmax > 0 and len(grads[True]) >= 14
    for k, v in grads[False].items():
        np.testing.assert_allclose(grads[True][k].cpu().numpy(), v.cpu().numpy(),
                                   atol=2e-5 * max(float(v.abs().max()), 1e-3 * gmax), rtol=1e-4, err_msg=k)


@pytest.mark.parametrize('rnn,n_markers', [(True, 12), (False, 6)])
def test_graphed_training_step_equals_eager(rnn, n_markers):
    """helpers/graphed.py: forward + backward captured in a HIP graph and replayed on new batches gives the losses and,
    after three Adam steps, the parameters of the eager loop (same seed, same batches, full-length windows)."""
    from em_pose_amd.data.data import SyntheticBatch
    from em_pose_amd.helpers.graphed import GraphedTrainStep
    model = H.small_model()
    vids = H.load_case('train_lgdrnn12_n2')['meta']['vertex_ids']   # twelve sensor vertices of the small test mesh
    B, F = 4, 8

    def make(seed):
        torch.manual_seed(seed)
        net = build_net(lgd_config(n_markers, rnn, 2, hidden=32, rnn_hidden=32), model, vids).train()
        params = [q for n, q in net.named_parameters() if not n.startswith('smpl.')]
        return net, params, torch.optim.Adam(params, lr=1e-3)

    def batch_of(seed, nb=B):
        w = synthetic.make_windows(nb, F, seed)
        g = torch.Generator().manual_seed(seed)
        w['marker_pos'] = torch.randn(nb, F, 36, generator=g).numpy()
        w['marker_oris'] = torch.randn(nb, F, 108, generator=g).numpy()
        b = SyntheticBatch(w, device=DEV)
        b.joints_gt = torch.randn(nb, F, 66, generator=g).to(DEV)
        return b
    batches = [batch_of(s) for s in (1, 2, 3)]

    net_e, params_e, opt_e = make(7)
    eager = []
    net_e.full_windows = True   # the captured step runs the LSTM unpacked, in pieces of 16 steps: same kernels here
    for b in batches:
        opt_e.zero_grad()
        _, vals = net_e.backward(b, net_e(b))
        opt_e.step()
        eager.append(vals)

    net_g, params_g, opt_g = make(7)
    step = GraphedTrainStep(net_g, opt_g, batches[0])
    for b, want in zip(batches, eager):
        vals = step(b)
        opt_g.step()
        torch.cuda.synchronize()
        for k, v in want.items():
            assert float(vals[k]) == pytest.approx(v, rel=1e-5, abs=1e-7), k
    for (k, p), q in zip(net_e.named_parameters(), net_g.parameters()):
        np.testing.assert_allclose(q.detach().cpu().numpy(), p.detach().cpu().numpy(), atol=2e-6, err_msg=k)
    with pytest.raises(ValueError):   # static shapes only
        step(batch_of(4, nb=B + 1))


def test_batched_streaming_equals_sequential():
    """Chunk c of all recordings as one ragged batch (state carried per row) gives the per-recording results of the
    one-recording-at-a-time driver."""
    from em_pose_amd.data.data import RealBatch, RealSample
    from em_pose_amd.data.transforms import NormalizeRealMarkers, NormalizeRoot, ToTensor
    from em_pose_amd.eval.helpers import evaluate_sequences, evaluate_sequences_batched
    case = H.load_case('lgdrnn6_n2')
    meta = case['meta']
    model = H.small_model()
    vids = [int(v) for v in meta['vertex_ids']]
    net = build_net(cfg_of(meta), model, vids, case['sd'])
    net.keep_history = False
    smpl = SMPLLayer(model).to(DEV)

    def sensors(poses, betas, o_r, o_t):
        n = poses.shape[0]
        p, o, _ = net.get_estimated_real_markers(gpu(poses), gpu(betas), gpu(o_r[:1]), gpu(o_t[:1]),
                                                 frames_per_window=n)
        return p.cpu().numpy(), o.cpu().numpy()
    batches = []
    for i, n in enumerate((300, 700, 256, 40)):
        d = synthetic.make_sequence(n, 50 + i, sensors, missing_rate=0.01)
        s = RealSample('r%d' % i, d['sensor_pos'], d['sensor_oris'], d['sensor_masks'].astype(np.float32),
                       d['smpl_poses'], d['smpl_shape'], d['smpl_trans'],
                       {'means': d['offset_means'], 'covs': d['offset_covs'], 'r': d['offset_r']})
        batches.append(NormalizeRoot()(RealBatch.from_sample_list([ToTensor()(NormalizeRealMarkers()(s))])))
    a_all, a_seq, a_frames = evaluate_sequences(net, batches, smpl, torch.device(DEV))
    b_all, b_seq, b_frames = evaluate_sequences_batched(net, batches, smpl, torch.device(DEV))
    assert a_frames == b_frames == 1296
    for (ida, ma), (idb, mb) in zip(a_seq, b_seq):
        assert ida == idb
        for k in ma:
            assert mb[k] == pytest.approx(ma[k], rel=1e-5, abs=1e-4), (ida, k)
    for k, v in a_all.get_metrics().items():
        assert b_all.get_metrics()[k] == pytest.approx(v, rel=1e-5, abs=1e-4)
    # the overall engine holds the rows in recording order, as the sequential driver accumulates them
    sa, sb = a_all.state(), b_all.state()
    for key in sa:
        assert sa[key].shape == sb[key].shape, key
        np.testing.assert_allclose(sb[key], sa[key], rtol=1e-4, atol=1e-4, err_msg=key)
    # page-locked fields (RealBatch.pin_memory, what a DataLoader with pin_memory=True hands over): same bits
    c_all, c_seq, c_frames = evaluate_sequences_batched(net, [b.pin_memory() for b in batches], smpl, torch.device(DEV))
    assert c_frames == b_frames and [i for i, _ in c_seq] == [i for i, _ in b_seq]
    for (_, mb), (_, mc) in zip(b_seq, c_seq):
        assert mb == mc
    sc = c_all.state()
    for key in sb:
        assert np.array_equal(sb[key], sc[key]), key


# ----------------------------------------------------------------------------------------------------------------------
# configuration flags and edge shapes (reference configuration.py:171-185; models.py:372-380,424-454)
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('flags', [
    dict(m_use_gradient=False),
    dict(m_average_shape=False),
    dict(m_skip_connections=True),
    dict(m_no_batch_norm=True),
    dict(m_num_iterations=0),
    dict(m_num_layers=1),
    dict(m_num_layers=3, m_skip_connections=True),
    dict(m_rnn_num_layers=1),
    dict(m_rnn_num_layers=3),
    dict(m_step_size=0.25, m_num_iterations=1),
], ids=lambda f: ','.join('%s=%s' % kv for kv in f.items()))
@pytest.mark.parametrize('rnn', [True, False], ids=['rnn', 'mlp'])
def test_configuration_flags_vs_oracle(flags, rnn):
    if not rnn and any(k.startswith('m_rnn') for k in flags):
        pytest.skip('LSTM flag without LSTM')
    model = H.small_model()
    vids = synthetic.small_vertex_ids(160)
    cfg = lgd_config(12, rnn, 2, hidden=32, rnn_hidden=32, **{k: v for k, v in flags.items() if k != 'm_num_iterations'})
    if 'm_num_iterations' in flags:
        cfg.m_num_iterations = flags['m_num_iterations']
    torch.manual_seed(5)
    net = create_model(cfg, SMPLLayer(model))
    g = torch.Generator().manual_seed(6)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
    net.vertex_ids = vids
    net = net.eval()
    bm = R.BodyModelTensors(model)
    tables = R.sensor_tables(model['f'], vids)

    def sensors(poses, betas, o_r, o_t):
        with torch.no_grad():
            p, o, _ = R.estimated_markers(bm, tables, vids, torch.from_numpy(poses), torch.from_numpy(betas),
                                          torch.from_numpy(o_r), torch.from_numpy(o_t))
        return p.numpy(), o.numpy()
    B, F = 3, 7
    w = synthetic.make_windows(B, F, 77, sensors)
    inp = H.oracle_inputs(w, sl=[7, 4, 1])
    sd = {k: v for k, v in net.state_dict().items() if not k.startswith('smpl.')}
    want, hist = R.ief_forward(sd, bm, tables, vids, inp, n_markers=12, N=cfg.m_num_iterations,
                               step_size=cfg.m_step_size, rnn_init=rnn, shape_avg=cfg.m_average_shape,
                               use_gradient=cfg.m_use_gradient, num_layers=cfg.m_num_layers,
                               batch_norm=not cfg.m_no_batch_norm, skip=cfg.m_skip_connections,
                               rnn_layers=cfg.m_rnn_num_layers)
    net = net.to(DEV)
    res = net.forward_tensors(*(inp[k].to(DEV) for k in ('marker_pos', 'marker_oris', 'offset_t', 'offset_r')),
                              seq_lengths=inp['seq_lengths'].to(DEV), keep_history=True)
    pose = res['pose'].cpu().numpy()
    np.testing.assert_allclose(pose[:, :, 3:], want['pose_hat'].numpy(), atol=ATOL)
    np.testing.assert_allclose(pose[:, :, :3], want['root_ori_hat'].numpy(), atol=ATOL)
    np.testing.assert_allclose(res['shape'].cpu().numpy(), want['shape_hat'].numpy(), atol=ATOL)
    np.testing.assert_allclose(res['joints'].cpu().numpy(), want['joints_hat'].numpy(), atol=ATOL)
    assert res['hist']['pose'].shape[0] == cfg.m_num_iterations + 1


@pytest.mark.parametrize('B,F', [(1, 1), (1, 2), (5, 3), (1, 257), (2, 1000)])
def test_edge_shapes_vs_oracle(B, F):
    """Tiny and long windows (the reference slices SMPL work at 1000 frames, smpl.py:124-144; evaluate_real feeds up
    to 256 frames per call): frame counts that are not multiples of any tile size."""
    case = H.load_case('lgdrnn12_n4_carry')
    model = H.small_model()
    vids = [int(v) for v in case['meta']['vertex_ids']]
    net = build_net(cfg_of(case['meta']), model, vids, case['sd'])
    bm = R.BodyModelTensors(model)
    tables = R.sensor_tables(model['f'], vids)

    def sensors(poses, betas, o_r, o_t):
        with torch.no_grad():
            p, o, _ = R.estimated_markers(bm, tables, vids, torch.from_numpy(poses), torch.from_numpy(betas),
                                          torch.from_numpy(o_r), torch.from_numpy(o_t))
        return p.numpy(), o.numpy()
    w = synthetic.make_windows(B, F, 3 + F, sensors)
    inp = H.oracle_inputs(w)
    want, _ = R.ief_forward(H.sd_to_torch(case['sd']), bm, tables, vids, inp, n_markers=12, N=4, rnn_init=True)
    res = net.forward_tensors(*(inp[k].to(DEV) for k in ('marker_pos', 'marker_oris', 'offset_t', 'offset_r')))
    pose = res['pose'].cpu().numpy()
    np.testing.assert_allclose(pose[:, :, 3:], want['pose_hat'].numpy(), atol=ATOL)
    np.testing.assert_allclose(res['shape'].cpu().numpy(), want['shape_hat'].numpy(), atol=ATOL)
    np.testing.assert_allclose(res['joints'].cpu().numpy(), want['joints_hat'].numpy(), atol=ATOL)
    np.testing.assert_allclose(res['state'][0].cpu().numpy(), _['rnn_state'][0].numpy(), atol=2e-5)


def test_invalid_calls_are_rejected():
    case = H.load_case('lgd12_n4')
    net = build_net(cfg_of(case['meta']), H.small_model(), case['meta']['vertex_ids'], case['sd'])
    lib = _lib.lib()
    h = net._ensure_handle(torch.device(DEV))
    io = _lib.LgdIO()
    io.B, io.F = 0, 4
    assert lib.empose_lgd_forward(h, C.byref(io), None, 0, None) == -1
    io.B = 2
    ws = torch.empty(16, dtype=torch.uint8, device=DEV)
    x = torch.zeros(2, 4, 108, device=DEV)
    io.marker_pos = io.marker_oris = io.offset_t = io.offset_r = _lib.dptr(x)
    io.pose_hat = io.shape_hat = io.joints_hat = _lib.dptr(x)
    assert lib.empose_lgd_forward(h, C.byref(io), _lib.dptr(ws), 16, None) == -3  # workspace too small
    assert b'workspace' in lib.empose_last_error()
    assert lib.empose_smpl_sensors_fwd_bwd(h, 5, 2, _lib.dptr(x), 66, _lib.dptr(x), 10, _lib.dptr(x), _lib.dptr(x), None,
                                           0, None, _lib.dptr(x), _lib.dptr(x), _lib.dptr(x), None, 0, None, 0,
                                           _lib.dptr(ws), 16, None) == -1  # T not a multiple of F


def test_smpl_degenerate_rotations(big_model):
    """Exactly-zero and tiny joint rotations (the smplx Rodrigues has no small-angle branch: angle = ||r + 1e-8||),
    large rotations close to pi, and extreme shapes: forward and residual gradient against the float64 blueprint."""
    T, F = 64, 8
    theta, beta, off_r, off_t, tgt, scale, _ = _smpl_case(big_model, CONST.VERTEX_IDS, T, F, 3, 12)
    theta[0:8, :] = 0.0                      # the rest pose, every joint exactly zero
    theta[8:16, 3:30] = 0.0                  # some joints exactly zero
    theta[16:24, :] *= 1e-4                  # tiny angles
    theta[24:32, 6:9] = np.array([3.1, 0.2, -0.1])  # close to pi
    theta[32:40, :] *= 4.0                   # large everywhere
    beta[40:48] = 4.0                        # far outside the usual +-2 range
    tab64 = TB.build_lgd_tables(big_model, CONST.VERTEX_IDS, dtype=np.float64)
    rep = lambda a: np.repeat(a, F, axis=0)
    idx = list(range(12))
    ref = A.smpl_sensors(tab64, theta, beta, rep(off_r), rep(off_t), tgt[:, :36].reshape(T, 12, 3),
                         tgt[:, 36:].reshape(T, 12, 3, 3), idx, scale)
    net = build_net(lgd_config(12, False, 1, hidden=32), big_model)
    handle = net._ensure_handle(torch.device(DEV))
    lib = _lib.lib()
    th, be, tg, o_r, o_t, sc = gpu(theta), gpu(beta), gpu(tgt), gpu(off_r), gpu(off_t), gpu(scale)
    pos, ori, joints = (torch.empty(T, n, device=DEV) for n in (36, 108, 66))
    g_th, g_be = torch.empty(T, 66, device=DEV), torch.empty(T, 10, device=DEV)
    nbytes = lib.empose_smpl_workspace_bytes(handle, T)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    _lib.check(lib.empose_smpl_sensors_fwd_bwd(handle, T, F, _lib.dptr(th), 66, _lib.dptr(be), 10, _lib.dptr(o_r),
                                               _lib.dptr(o_t), _lib.dptr(tg), 144, _lib.dptr(sc), _lib.dptr(pos),
                                               _lib.dptr(ori), _lib.dptr(joints), _lib.dptr(g_th), 66, _lib.dptr(g_be),
                                               10, _lib.dptr(ws), nbytes, _lib.current_stream()))
    torch.cuda.synchronize()
    assert torch.isfinite(g_th).all() and torch.isfinite(pos).all()
    np.testing.assert_allclose(pos.cpu().numpy().reshape(T, 12, 3), ref['pos'], atol=2e-5)
    np.testing.assert_allclose(ori.cpu().numpy().reshape(T, 12, 3, 3), ref['ori'], atol=5e-5)
    np.testing.assert_allclose(joints.cpu().numpy().reshape(T, 22, 3), ref['joints'], atol=2e-5)
    # the rest pose reproduces the regressed template joints exactly
    rest = big_model['J_regressor'][:22].astype(np.float64) @ (
        big_model['v_template'].astype(np.float64) +
        np.einsum('vkl,l->vk', big_model['shapedirs'][:, :, :10].astype(np.float64), beta[0]))
    np.testing.assert_allclose(joints.cpu().numpy().reshape(T, 22, 3)[0], rest, atol=2e-6)
    gmax = np.abs(ref['g_theta']).max()
    np.testing.assert_allclose(g_th.cpu().numpy(), ref['g_theta'], atol=5e-4 * gmax, rtol=2e-3)
    np.testing.assert_allclose(g_be.cpu().numpy(), ref['g_beta'], atol=5e-4 * np.abs(ref['g_beta']).max(), rtol=2e-3)


def test_device_metrics_vs_reference_vectors_and_numpy(big_model):
    """empose_metrics_rows: MPJPE / PA-MPJPE against the numbers recorded from the reference's MetricsEngine, and all
    three metrics (incl. the global joint-angle error) against the host NumPy implementation."""
    import os
    from em_pose_amd.eval.metrics import (MetricsEngine, geodesic_degrees, local_to_global_rotations,
                                          procrustes_align)
    z = np.load(os.path.join(H.GOLDEN, 'components.npz'))
    me = MetricsEngine(None)
    me.compute_joint_dist(gpu(z['me_joints']), gpu(z['me_joints_hat']), gpu(z['me_len'], torch.int64),
                          gpu(z['me_mask']))
    got = me.get_metrics()
    assert got['MPJPE [mm]'] == pytest.approx(float(z['me_MPJPE']), rel=1e-5)
    assert got['MPJPE STD'] == pytest.approx(float(z['me_MPJPE_STD']), rel=1e-5)
    assert got['PA-MPJPE [mm]'] == pytest.approx(float(z['me_PA-MPJPE']), rel=1e-5)
    assert got['PA-MPJPE STD'] == pytest.approx(float(z['me_PA-MPJPE_STD']), rel=1e-5)

    # full compute(): device path vs host path on the same predictions
    smpl = SMPLLayer(big_model).to(DEV)
    rng = np.random.default_rng(4)
    n, f = 3, 50
    pose = rng.normal(0, 0.3, size=(n, f, 63)).astype(np.float32)
    pose[0, :5] *= 1e-3                      # tiny angles (the reference's exp map clamps the angle at 1e-2)
    pose_hat = pose + rng.normal(0, 0.1, size=pose.shape).astype(np.float32)
    root, root_hat = [rng.normal(0, 0.3, size=(n, f, 3)).astype(np.float32) for _ in range(2)]
    shape, shape_hat = [rng.normal(0, 1, size=(n, 10)).astype(np.float32) for _ in range(2)]
    lens = np.array([50, 31, 7])
    dev_me = MetricsEngine(smpl)
    dev_me.compute(gpu(pose), gpu(shape), gpu(pose_hat), gpu(shape_hat), gpu(lens, torch.int64), gpu(root),
                   gpu(root_hat))
    rows = dev_me.state()
    assert rows['eucl'].shape == (88, 22) and rows['angle'].shape == (88, 21)
    valid = np.arange(f)[None, :] < lens[:, None]
    P, Ph = pose[valid].astype(np.float64), pose_hat[valid].astype(np.float64)
    zeros = np.zeros((P.shape[0], 3))
    g = local_to_global_rotations(np.concatenate([zeros, P], -1), CONST.SMPL_PARENTS)[:, 1:]
    gh = local_to_global_rotations(np.concatenate([zeros, Ph], -1), CONST.SMPL_PARENTS)[:, 1:]
    np.testing.assert_allclose(rows['angle'], geodesic_degrees(g, gh), atol=2e-3)  # clamp differs below 0.01 rad only
    bm = R.BodyModelTensors(big_model)
    rep = lambda s: torch.from_numpy(np.repeat(s[:, None], f, axis=1)[valid])
    _, j = R.smpl_fk(bm, torch.from_numpy(pose[valid]), rep(shape), torch.from_numpy(root[valid]))
    _, jh = R.smpl_fk(bm, torch.from_numpy(pose_hat[valid]), rep(shape_hat), torch.from_numpy(root_hat[valid]))
    j, jh = j[:, :22].numpy().astype(np.float64), jh[:, :22].numpy().astype(np.float64)
    np.testing.assert_allclose(rows['eucl'], np.linalg.norm(j - jh, axis=-1), atol=5e-6)
    np.testing.assert_allclose(rows['eucl_pa'], np.linalg.norm(j - procrustes_align(j, jh), axis=-1), atol=5e-6)


def test_amass_batch_through_preprocessing_and_a_training_step():
    """The training-side containers end to end: AMASSSample -> AMASSBatch -> NormalizeRoot/SMPLFK/SampleMarkersWithOffsets
    (reference transforms.py:23-48) -> one forward + backward of LGD-RNN in training mode."""
    from em_pose_amd.data.data import AMASSBatch, AMASSSample
    from em_pose_amd.data.transforms import ToTensor, get_end_to_end_preprocess_fn
    case = H.load_case('train_lgdrnn12_n2')
    model, vids = H.small_model(), [int(v) for v in case['meta']['vertex_ids']]
    smpl = SMPLLayer(model).to(DEV)
    rng = np.random.default_rng(8)
    samples = []
    for i, n in enumerate((6, 4)):
        smp = AMASSSample('s%d' % i, rng.normal(0, 0.2, size=(n, 66)).astype(np.float32),
                          rng.normal(0, 1, size=10).astype(np.float32), rng.normal(0, 1, size=(n, 3)).astype(np.float32), 60.0)
        samples.append(ToTensor()(smp))
    batch = AMASSBatch.from_sample_list(samples).to_gpu(torch.device(DEV))
    offsets = {'means': rng.normal(0, 0.02, size=(12, 3)).astype(np.float32), 'covs': None,
               'r': np.tile(np.eye(3, dtype=np.float32), (12, 1, 1)), 'vertex_ids': np.asarray(vids)}
    batch = get_end_to_end_preprocess_fn(cfg_of(case['meta']), smpl, [offsets])(batch)
    assert batch.marker_pos_synth.shape == (2, 6, 36) and batch.joints_gt.shape == (2, 6, 66)
    net = build_net(cfg_of(case['meta']), model, vids, case['sd']).train()
    net.zero_grad()
    total, vals = net.backward(batch, net(batch))
    assert np.isfinite(vals['total_loss']) and vals['total_loss'] > 0
    grads = [p.grad for n_, p in net.named_parameters() if not n_.startswith('smpl.') and p.grad is not None]
    assert len(grads) >= 14 and all(bool(torch.isfinite(g).all()) for g in grads)


def test_validation_loop_evaluate():
    """eval/helpers.py::evaluate (reference eval/helpers.py:51-111): losses averaged over the samples of a loader, metrics
    accumulated for every valid frame; plus MetricsEngine.compute_angle_dist on raw joint angles."""
    from em_pose_amd.data.data import AMASSBatch, AMASSSample
    from em_pose_amd.data.transforms import ToTensor, get_end_to_end_preprocess_fn
    from em_pose_amd.eval.helpers import evaluate
    from em_pose_amd.eval.metrics import MetricsEngine, geodesic_degrees, rotvec_to_matrix
    case = H.load_case('train_lgdrnn12_n2')
    model, vids = H.small_model(), [int(v) for v in case['meta']['vertex_ids']]
    smpl = SMPLLayer(model).to(DEV)
    net = build_net(cfg_of(case['meta']), model, vids, case['sd'])
    rng = np.random.default_rng(9)
    offsets = {'means': rng.normal(0, 0.02, size=(12, 3)).astype(np.float32), 'covs': None,
               'r': np.tile(np.eye(3, dtype=np.float32), (12, 1, 1)), 'vertex_ids': np.asarray(vids)}
    fn = get_end_to_end_preprocess_fn(cfg_of(case['meta']), smpl, [offsets])

    def loader():
        r = np.random.default_rng(10)
        for lens in ((6, 4), (5,)):
            yield AMASSBatch.from_sample_list([ToTensor()(AMASSSample(
                's', r.normal(0, 0.2, size=(n, 66)).astype(np.float32), r.normal(0, 1, size=10).astype(np.float32),
                np.zeros((n, 3), np.float32), 60.0)) for n in lens])
    me = MetricsEngine(smpl)
    losses = evaluate(loader(), net, fn, me, device=torch.device(DEV))
    assert set(losses) >= {'pose', 'shape', 'reconstruction', 'fk', 'total_loss'} and np.isfinite(losses['total_loss'])
    assert np.concatenate(me.eucl_dists).shape == (15, 22) and net.keep_history in (True, False)
    # the same number by hand: per-batch loss weighted by its batch size
    net.keep_history = True
    want, n = 0.0, 0
    with torch.no_grad():
        for ab in loader():
            b = fn(ab.to_gpu(torch.device(DEV)), mode='all')
            want += net.backward(b, net(b))[1]['total_loss'] * b.batch_size
            n += b.batch_size
    assert losses['total_loss'] == pytest.approx(want / n, rel=1e-5)
    # joint-angle metric on the angles as given
    p, ph = rng.normal(0, 0.5, size=(2, 3, 63)), rng.normal(0, 0.5, size=(2, 3, 63))
    me2 = MetricsEngine(None)
    me2.compute_angle_dist(torch.from_numpy(p), torch.from_numpy(ph), seq_lengths=torch.tensor([3, 2]))
    rows = np.concatenate(me2.angle_diffs)
    valid = np.array([[1, 1, 1], [1, 1, 0]], bool)
    np.testing.assert_allclose(rows, geodesic_degrees(rotvec_to_matrix(p[valid].reshape(5, 21, 3)),
                                                      rotvec_to_matrix(ph[valid].reshape(5, 21, 3))), atol=1e-9)


def test_vertex_normals_api():
    """SMPLLayer.vertex_normals / VirtualMarkerHelper.get_vertex_normals (reference smpl.py:69-79,
    virtual_sensors.py:77-83): mean of the incident faces' un-normalised normals."""
    model = H.small_model()
    smpl = SMPLLayer(model).to(DEV)
    rng = np.random.default_rng(12)
    v = rng.normal(size=(3, model['v_template'].shape[0], 3)).astype(np.float32)
    f = np.asarray(model['f'], dtype=np.int64)
    fn = np.cross(v[:, f[:, 1]] - v[:, f[:, 0]], v[:, f[:, 2]] - v[:, f[:, 0]])        # (3, n_faces, 3)
    want = np.zeros_like(v)
    deg = np.zeros(v.shape[1])
    for k, tri in enumerate(f):
        for vid in tri:
            want[:, vid] += fn[:, k]
            deg[vid] += 1
    want /= np.maximum(deg, 1)[None, :, None]
    got = smpl.vertex_normals(gpu(v)).cpu().numpy()
    used = deg > 0
    np.testing.assert_allclose(got[:, used], want[:, used], atol=2e-5)
    ids = [5, 17, 100]
    np.testing.assert_allclose(smpl.vertex_normals(gpu(v), ids).cpu().numpy(), want[:, ids], atol=2e-5)
    from em_pose_amd.data.virtual_sensors import VirtualMarkerHelper
    helper = VirtualMarkerHelper(smpl)
    np.testing.assert_allclose(helper.get_vertex_normals(gpu(v), ids).cpu().numpy(), want[:, ids], atol=2e-5)


def test_ground_truth_preprocessing_round_trip(big_model):
    """SMPLFK + SampleMarkersWithOffsets: sensors sampled from the full ground-truth mesh with offsets
    equal what the LGD sub-mesh path predicts for the same pose/shape/offsets (two independent HIP routes)."""
    from em_pose_amd.data.data import SyntheticBatch
    from em_pose_amd.data.transforms import SMPLFK, SampleMarkersWithOffsets
    smpl = SMPLLayer(big_model).to(DEV)
    w = synthetic.make_windows(3, 5, 8)
    w['marker_pos'] = np.zeros((3, 5, 36), np.float32)
    w['marker_oris'] = np.zeros((3, 5, 108), np.float32)
    batch = SyntheticBatch(w, device=DEV)
    sets = [{'means': w['offset_t'][i], 'r': w['offset_r'][i], 'vertex_ids': np.asarray(CONST.VERTEX_IDS)}
            for i in range(3)]
    tr = SampleMarkersWithOffsets(smpl, sets)
    batch = tr(SMPLFK(smpl)(batch))
    assert batch.vertices.shape == (3, 5, 6890 * 3) and batch.joints_gt.shape == (3, 5, 66)
    net = build_net(lgd_config(12, False, 1, hidden=32), big_model)
    pos, ori, joints = net.get_estimated_real_markers(batch.poses.reshape(15, 66),
                                                      batch.shapes[:, None].expand(3, 5, 10).reshape(15, 10),
                                                      batch.offset_r_augmented, batch.offset_t_augmented,
                                                      frames_per_window=5)
    np.testing.assert_allclose(batch.marker_pos_synth.cpu().numpy().reshape(15, 12, 3), pos.cpu().numpy(), atol=2e-5)
    np.testing.assert_allclose(batch.marker_ori_synth.cpu().numpy().reshape(15, 12, 3, 3), ori.cpu().numpy(), atol=5e-5)
    np.testing.assert_allclose(batch.joints_gt.cpu().numpy().reshape(15, 22, 3), joints.cpu().numpy(), atol=2e-5)
    # the reference's factory composes the same three transforms (transforms.py:23-48)
    from em_pose_amd.data.transforms import get_end_to_end_preprocess_fn
    fn = get_end_to_end_preprocess_fn(lgd_config(12, False, 1, hidden=32), smpl, sets)
    again = fn(SyntheticBatch(w, device=DEV), mode='after_normalize')
    np.testing.assert_allclose(again.marker_pos_synth.cpu().numpy(), batch.marker_pos_synth.cpu().numpy(), atol=1e-6)
    # the training-time noise levels of the reference (transforms.py:176-211)
    for s_ in sets:
        s_['covs'] = np.tile(np.eye(3, dtype=np.float32) * 1e-4, (12, 1, 1))
    det_pos, det_ori = batch.marker_pos_synth.clone(), batch.marker_ori_synth.clone()
    torch.manual_seed(0)
    b0 = SampleMarkersWithOffsets(smpl, sets, noise_level=0)(batch)
    d0 = (b0.marker_pos_synth - b0.marker_pos_vertex).reshape(3, 5, 12, 3)
    local0 = torch.matmul(b0.marker_ori_vertex.reshape(3, 5, 12, 3, 3).transpose(-1, -2), d0.unsqueeze(-1)).squeeze(-1)
    np.testing.assert_allclose(local0[:, 1:].cpu().numpy(), local0[:, :1].expand(3, 4, 12, 3).cpu().numpy(), atol=1e-5)
    assert float((local0[:, 0] - b0.offset_t_augmented).abs().max()) < 0.1 and float((local0[:, 0] - b0.offset_t_augmented).abs().max()) > 1e-4
    b1 = SampleMarkersWithOffsets(smpl, sets, noise_level=1)(batch)
    d1 = (b1.marker_pos_synth - b1.marker_pos_vertex).reshape(3, 5, 12, 3)
    local1 = torch.matmul(b1.marker_ori_vertex.reshape(3, 5, 12, 3, 3).transpose(-1, -2), d1.unsqueeze(-1)).squeeze(-1)
    assert float((local1[:, 1] - local1[:, 0]).abs().max()) > 1e-4        # a new draw every frame
    b2 = SampleMarkersWithOffsets(smpl, sets, noise_level=2)(batch)
    np.testing.assert_allclose(b2.marker_pos_synth.cpu().numpy(), b2.marker_pos_vertex.cpu().numpy(), atol=0)
    b3 = SampleMarkersWithOffsets(smpl, sets, noise_level=3)(batch)
    np.testing.assert_allclose(b3.marker_ori_synth.cpu().numpy(), b3.marker_ori_vertex.cpu().numpy(), atol=1e-6)
    np.testing.assert_allclose(b3.offset_r_augmented.cpu().numpy(), np.broadcast_to(np.eye(3), (3, 12, 3, 3)), atol=0)
    with pytest.raises(ValueError):
        SampleMarkersWithOffsets(smpl, sets, noise_level=4)
    assert det_pos.shape == b3.marker_pos_synth.shape and det_ori.shape == b3.marker_ori_synth.shape


# ----------------------------------------------------------------------------------------------------------------------
# Training backward building blocks (BASELINE configs[4]): hand-written kernels against torch.autograd
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('M,N,K', [(384, 512, 512), (8192, 512, 296), (1000, 66, 512), (50, 10, 512), (4096, 2048, 144),
                                   (7, 5, 3), (130, 200, 320), (24576, 512, 512), (20000, 515, 500)])
def test_gemm_atb_vs_float64(M, N, K):
    """C = A^T B (+ column sums of A): the weight / bias gradient of a linear layer.  The last two shapes are long
    reductions (twice the split, 16-row chunks, two workgroups per CU), the last one with edge tiles and a ragged end."""
    rng = np.random.default_rng(M + N)
    lda, ldb, ldc = N + 3, K + 5, K + 2
    a = rng.normal(size=(M, lda)).astype(np.float32)
    b = rng.normal(size=(M, ldb)).astype(np.float32)
    want = a[:, :N].astype(np.float64).T @ b[:, :K].astype(np.float64)
    A_, B_ = gpu(a), gpu(b)
    Cm = torch.full((N, ldc), -7.0, device=DEV)
    bias = torch.full((N,), -7.0, device=DEV)
    lib = _lib.lib()
    nbytes = lib.empose_gemm_atb_workspace_bytes(M, N, K)
    ws = torch.empty(max(nbytes, 4), dtype=torch.uint8, device=DEV)
    for _ in range(2):   # twice: bitwise reproducible
        _lib.check(lib.empose_gemm_atb_f32(M, N, K, _lib.dptr(A_), lda, _lib.dptr(B_), ldb, _lib.dptr(Cm), ldc,
                                           _lib.dptr(bias), _lib.dptr(ws), ws.numel(), _lib.current_stream()))
        torch.cuda.synchronize()
        got, gb = Cm.cpu().numpy(), bias.cpu().numpy()
        np.testing.assert_allclose(got[:, :K], want, atol=2e-5 * np.sqrt(M), rtol=1e-5)
        np.testing.assert_allclose(gb, a[:, :N].astype(np.float64).sum(0), atol=2e-5 * np.sqrt(M))
        assert (got[:, K:] == -7.0).all()
        if _ == 0:
            first = got.copy()
    assert np.array_equal(first, got)


def test_transpose_f32():
    rng = np.random.default_rng(1)
    for rows, cols in ((2048, 512), (66, 512), (33, 7)):
        a = rng.normal(size=(rows, cols + 4)).astype(np.float32)
        A_ = gpu(a)
        out = torch.zeros(cols, rows + 2, device=DEV)
        _lib.check(_lib.lib().empose_transpose_f32(rows, cols, _lib.dptr(A_), cols + 4, _lib.dptr(out), rows + 2,
                                                   _lib.current_stream()))
        assert np.array_equal(out.cpu().numpy()[:, :rows], a[:, :cols].T)


@pytest.mark.parametrize('M,K,N', [(8192, 296, 512), (2048, 512, 512), (3000, 512, 66), (1536, 512, 10)])
def test_linear_train_large_batch_vs_torch_autograd(M, K, N):
    from em_pose_amd.nn.layers import linear_train, _HipLinearLargeFn  # noqa: F401
    torch.manual_seed(M + N)
    lin = torch.nn.Linear(K, N).to(DEV)
    x = torch.randn(M, K, device=DEV, requires_grad=True)
    dy = torch.randn(M, N, device=DEV)
    y = linear_train(x, lin)
    y.backward(dy)
    got = [y.detach().clone(), x.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone()]
    x.grad = None
    lin.zero_grad()
    y2 = torch.nn.functional.linear(x.double(), lin.weight.double(), lin.bias.double())
    y2.backward(dy.double())
    want = [y2.detach(), x.grad.double(), None, None]
    np.testing.assert_allclose(got[0].cpu().numpy(), want[0].cpu().numpy(), atol=3e-5 * np.sqrt(K / 32))
    np.testing.assert_allclose(got[1].cpu().numpy(), want[1].cpu().numpy(), atol=3e-5 * np.sqrt(N / 32))
    dw = dy.double().t() @ x.detach().double()
    np.testing.assert_allclose(got[2].cpu().numpy(), dw.cpu().numpy(), atol=3e-5 * np.sqrt(M))
    np.testing.assert_allclose(got[3].cpu().numpy(), dy.double().sum(0).cpu().numpy(), atol=3e-5 * np.sqrt(M))


@pytest.mark.parametrize('B,F,K,H,L,ragged,carry', [(12, 32, 144, 512, 2, False, False), (5, 9, 72, 64, 2, True, True),
                                                    (40, 16, 144, 128, 1, True, False), (300, 8, 144, 256, 2, False, True),
                                                    (3, 4, 8, 16, 3, True, False),
                                                    (256, 5, 144, 512, 2, True, True),    # K-split recurrent products
                                                    (12, 6, 144, 256, 2, True, True),     # wavefront, matrix-vector form
                                                    (130, 3, 16, 128, 2, True, False)])   # wavefront, K-split form
def test_lstm_training_forward_backward_vs_torch(B, F, K, H, L, ragged, carry):
    """empose_lstm_train_fwd/bwd (all three step kernels: B <= 16, <= 256, larger) against torch.nn.LSTM over packed
    sequences with autograd, in float64 on the CPU: outputs, final state, dx and every parameter gradient."""
    from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence
    from em_pose_amd.nn.layers import _LstmTrainFn
    torch.manual_seed(B * 7 + F)
    ref = torch.nn.LSTM(K, H, L).double()
    x = torch.randn(B, F, K, dtype=torch.float64)
    lens = torch.randint(1, F + 1, (B,)) if ragged else torch.full((B,), F)
    lens[0] = F
    state = (0.5 * torch.randn(L, B, H, dtype=torch.float64), 0.5 * torch.randn(L, B, H, dtype=torch.float64)) if carry else None
    if carry:   # (round 6: the cotangents of a given initial state -- what a learned one is trained with)
        state = tuple(t.requires_grad_(True) for t in state)
    dy = torch.randn(B, F, H, dtype=torch.float64)
    xr = x.clone().requires_grad_(True)
    packed = pack_padded_sequence(xr, lens, batch_first=True, enforce_sorted=False)
    out, (hn, cn) = ref(packed, state)
    out, _ = pad_packed_sequence(out, batch_first=True, total_length=F)
    (out * dy).sum().backward()
    weights = [getattr(ref, '%s_l%d' % (n, l)) for l in range(L) for n in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh')]
    wg = [w.detach().float().to(DEV).requires_grad_(True) for w in weights]
    xg = x.float().to(DEV).requires_grad_(True)
    h0 = c0 = None
    if carry:
        h0, c0 = [t.detach().float().to(DEV).requires_grad_(True) for t in state]
    y, h_n, c_n = _LstmTrainFn.apply(xg, lens.to(DEV, torch.int32) if ragged else None, h0, c0, L, *wg)
    (y * dy.float().to(DEV)).sum().backward()
    torch.cuda.synchronize()
    tol = 2e-5
    np.testing.assert_allclose(y.detach().cpu().numpy(), out.detach().numpy(), atol=tol)
    np.testing.assert_allclose(h_n.cpu().numpy(), hn.detach().numpy(), atol=tol)
    np.testing.assert_allclose(c_n.cpu().numpy(), cn.detach().numpy(), atol=tol)
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xr.grad.numpy(), atol=5e-5)
    for g_, w in zip(wg, weights):
        scale = max(1.0, float(w.grad.abs().max()))
        np.testing.assert_allclose(g_.grad.cpu().numpy(), w.grad.numpy(), atol=1e-4 * scale)
    if carry:
        np.testing.assert_allclose(h0.grad.cpu().numpy(), state[0].grad.numpy(), atol=5e-5)
        np.testing.assert_allclose(c0.grad.cpu().numpy(), state[1].grad.numpy(), atol=5e-5)


def test_lstm_training_forward_whole_sequence_kernel_equals_step_launches():
    """Small batches run the training forward as one cooperative launch too (lstm_persist_kernel now stores the gates /
    cell states / incoming hidden states that back-propagation through time reads): same bits as the step-by-step
    launches (option lstm_persist = 0) for outputs, final state and every gradient, ragged lengths and carried state."""
    from em_pose_amd.nn.layers import _LstmTrainFn
    torch.manual_seed(11)
    B, F, K, H, L = 12, 32, 144, 512, 2
    x = torch.randn(B, F, K, device=DEV)
    lens = torch.randint(1, F + 1, (B,), dtype=torch.int32)
    lens[3] = F
    h0, c0 = 0.5 * torch.randn(L, B, H, device=DEV), 0.5 * torch.randn(L, B, H, device=DEV)
    dy = torch.randn(B, F, H, device=DEV)
    ws = [0.05 * torch.randn(*shape, device=DEV) for l in range(L)
          for shape in ((4 * H, K if l == 0 else H), (4 * H, H), (4 * H,), (4 * H,))]
    lib = _lib.lib()
    _set_option(b'lstm_fewrows', 0)     # (round 5: the two kernels that share their bits; lstm_fewrows_kernel: test_hip_round5.py)
    res = {}
    for mode in (1, 0):
        _lib.check(lib.empose_set_option(b'lstm_persist', mode))
        try:
            wg = [w.clone().requires_grad_(True) for w in ws]
            xg = x.clone().requires_grad_(True)
            y, h_n, c_n = _LstmTrainFn.apply(xg, lens.to(DEV), h0, c0, L, *wg)
            (y * dy).sum().backward()
            torch.cuda.synchronize()
            res[mode] = [y.detach(), h_n, c_n, xg.grad] + [w.grad for w in wg]
        finally:
            _lib.check(lib.empose_set_option(b'lstm_persist', 1))
    for a, b in zip(res[1], res[0]):
        assert torch.equal(a, b)
    assert torch.isfinite(res[1][0]).all()


def test_hip_adam_equals_torch_adam():
    """empose_adam_step (one launch over all tensors) against torch.optim.Adam, three steps."""
    from em_pose_amd.helpers.optim import HipAdam
    torch.manual_seed(3)
    shapes = [(512, 512), (66,), (5000,), (1,), (2048, 144)]
    ours = [torch.randn(*s, device=DEV).requires_grad_(True) for s in shapes]
    ref = [p.detach().clone().requires_grad_(True) for p in ours]
    oa, ob = HipAdam(ours, lr=5e-4), torch.optim.Adam(ref, lr=5e-4)
    for step in range(3):
        for p, q in zip(ours, ref):
            g = torch.randn_like(p)
            p.grad, q.grad = g.clone(), g.clone()
        oa.step()
        ob.step()
    torch.cuda.synchronize()
    for p, q in zip(ours, ref):
        np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().cpu().numpy(), atol=1e-6, rtol=1e-6)
