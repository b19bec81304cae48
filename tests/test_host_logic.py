"""Host-side logic of the package (CPU): configuration, model construction, packed tables, batch containers."""
import json
import os

import numpy as np
import pytest
import torch

from em_pose_amd import synthetic
from em_pose_amd.bodymodels import tables as TB
from em_pose_amd.bodymodels.smpl import SMPLLayer
from em_pose_amd.data.data import RealBatch
from em_pose_amd.helpers.configuration import CONSTANTS as C
from em_pose_amd.helpers.configuration import Configuration, lgd_config
from em_pose_amd.nn.models import create_model, mask_from_seq_lengths, reconstruction_loss
from tests import helpers as H


def test_known_answers_of_released_configurations():
    """Parameter counts / model names recorded from the reference (tests/golden/known_answers.json, README.md:228-229)."""
    known = json.load(open(os.path.join(H.GOLDEN, 'known_answers.json')))
    smpl = SMPLLayer(H.small_model())
    for tag, cfg in (('lgd_rnn_6_N2', lgd_config(6, True, 2, lr=0.0005)),
                     ('lgd_rnn_12_N4', lgd_config(12, True, 4)),
                     ('lgd_12_N4', lgd_config(12, False, 4))):
        net = create_model(cfg, smpl)
        n = sum(p.numel() for k, p in net.named_parameters() if not k.startswith('smpl.'))
        assert n == known[tag]['params_without_bodymodel']
        assert net.model_name() == known[tag]['model_name']
    net = create_model(lgd_config(6, True, 2, lr=0.0005), smpl)
    assert sum(p.numel() for p in net.parameters()) == 5721419  # reference README.md:228


def test_state_dict_keys_match_reference_checkpoints():
    smpl = SMPLLayer(H.small_model())
    for name, rnn in (('lgdrnn12_n4_carry', True), ('lgd12_n4', False)):
        case = H.load_case(name)
        net = create_model(lgd_config(12, rnn, 4, hidden=32, rnn_hidden=32), smpl)
        mine = {k for k in net.state_dict() if not k.startswith('smpl.')}
        assert mine == set(case['sd'])
        for k, v in net.state_dict().items():
            if not k.startswith('smpl.'):
                assert tuple(v.shape) == case['sd'][k].shape, k
    keys = {k for k in net.state_dict() if k.startswith('smpl.bm.')}
    assert {'smpl.bm.f', 'smpl.bm.v_template', 'smpl.bm.shapedirs', 'smpl.bm.posedirs', 'smpl.bm.J_regressor',
            'smpl.bm.weights'} <= keys
    assert net.state_dict()['smpl.bm.posedirs'].shape == (459, 160 * 3)


def test_configuration_roundtrip(tmp_path):
    cfg = lgd_config(6, True, 2)
    p = tmp_path / 'config.json'
    cfg.to_json(str(p))
    back = Configuration.from_json(str(p))
    assert vars(back) == vars(cfg)
    cli = Configuration.parse_cmd(['--m_type', 'ief', '--n_markers', '6', '--m_rnn_init', '--use_marker_pos'])
    assert cli.m_type == 'ief' and cli.n_markers == 6 and cli.m_rnn_init and cli.m_step_size == 0.1
    assert C.S_CONFIG_6 == [0, 1, 2, 6, 7, 11] and len(C.VERTEX_IDS) == 12 and len(C.SMPL_PARENTS) == 22


def test_tables_match_reference_topology_vectors():
    z = np.load(os.path.join(H.GOLDEN, 'components.npz'))
    model = H.small_model()
    vids = synthetic.small_vertex_ids(160)
    sub_faces, vf_sub, helpers = TB.sensor_topology(model['f'], vids)
    assert (sub_faces == z['vs_sub_faces']).all() and (vf_sub == z['vs_sub_vertex_faces']).all()
    assert (helpers == z['vs_helpers']).all()
    tab = TB.build_lgd_tables(model, vids)
    # every sensor patch refers to needed vertices only, helper is part of the ring
    assert tab['s_faces'].max() < tab['nv'] and (tab['s_deg'] == 6).all()
    for m in range(12):
        ring = set(tab['s_faces'][m, :tab['s_deg'][m]].reshape(-1).tolist())
        assert tab['s_center'][m] in ring and tab['s_helper'][m] in ring
    assert tab['needed'][tab['s_center']].tolist() == vids
    # folded skinning weights stay convex; CSR by bone is the transpose of the per-vertex table
    np.testing.assert_allclose(tab['skin_w'].sum(1), 1.0, atol=1e-6)
    dense = np.zeros((tab['nv'], 22))
    for s in range(tab['nv']):
        for k in range(tab['kb']):
            dense[s, tab['skin_idx'][s, k]] += tab['skin_w'][s, k]
    dense2 = np.zeros_like(dense)
    for b in range(22):
        for q in range(tab['bone_ptr'][b], tab['bone_ptr'][b + 1]):
            dense2[tab['bone_vert'][q], b] = tab['bone_w'][q]
    np.testing.assert_allclose(dense, dense2, atol=0)
    # tree walks
    assert tab['path'][tab['path_ptr'][20]:tab['path_ptr'][21]].tolist() == [0, 3, 6, 9, 13, 16, 18, 20]
    assert sorted(tab['sub'][tab['sub_ptr'][16]:tab['sub_ptr'][17]].tolist()) == [16, 18, 20]
    assert tab['sub_ptr'][1] == 22  # the root's subtree is everything


def test_vertex_faces_order_is_trimesh_sparse_product_order():
    """`Trimesh.vertex_faces` (trimesh==3.9.32) fills its rows from `faces_sparse.dot(identity).nonzero()[1]`; the
    stand-in oracle/refstubs/trimesh evaluates that scipy expression, the product and the oracle state the resulting
    order (descending face id) directly.  The helper vertex of every sensor frame hangs on row[0] of this table
    (reference virtual_sensors.py:55)."""
    import importlib.util
    from oracle import torch_ref as R
    spec = importlib.util.spec_from_file_location(
        'trimesh_standin', os.path.join(os.path.dirname(H.GOLDEN), '..', 'oracle', 'refstubs', 'trimesh', '__init__.py'))
    tm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tm)
    for model in (H.small_model(), synthetic.make_model()):
        faces = np.asarray(model['f'], dtype=np.int64)
        n = int(faces.max()) + 1
        want = tm.Trimesh(np.zeros((n, 3)), faces, process=False).vertex_faces
        assert (TB.vertex_faces_table(faces, n) == want).all()
        assert (R.vertex_faces_table(faces, n) == want).all()
        live = want[:, 0] >= 0
        assert (want[live, 0] == np.where(want[live] >= 0, want[live], -1).max(axis=1)).all()   # row[0] = largest face id
    # a ragged mesh (vertex degrees 1..4) keeps the -1 padding on the right
    faces = np.array([[0, 1, 2], [0, 2, 3], [0, 3, 4], [0, 4, 5], [5, 6, 7]])
    want = tm.Trimesh(np.zeros((8, 3)), faces, process=False).vertex_faces
    assert (TB.vertex_faces_table(faces, 8) == want).all() and want[0].tolist() == [3, 2, 1, 0]


def test_loss_helpers_match_reference_vectors():
    z = np.load(os.path.join(H.GOLDEN, 'components.npz'))
    gt, hat = torch.from_numpy(z['rl_gt']), torch.from_numpy(z['rl_hat'])
    sl, mm = torch.from_numpy(z['rl_len']), torch.from_numpy(z['rl_mask'])
    np.testing.assert_allclose(reconstruction_loss(gt, hat, sl, mm).numpy(), z['rl_full'], rtol=1e-6)
    np.testing.assert_allclose(reconstruction_loss(gt, hat).numpy(), z['rl_plain'], rtol=1e-6)
    assert (mask_from_seq_lengths(sl).numpy() == z['mask_from_len']).all()


def test_real_batch_suppresses_missing_sensors():
    B, F = 2, 5
    masks = torch.ones(B, F, 12)
    masks[0, 2, 3] = 0
    b = RealBatch([0, 1], torch.tensor([5, 5]), torch.zeros(B, F, 66), torch.zeros(B, 10), torch.zeros(B, F, 3),
                  torch.ones(B, F, 36), torch.ones(B, F, 108), masks)
    inp = b.get_inputs(sf=1, ef=4)
    assert inp['marker_pos'].shape == (B, 3, 36) and inp['marker_masks'].shape == (B, 3, 12)
    assert inp['marker_pos'].reshape(B, 3, 12, 3)[0, 1, 3].abs().sum() == 0
    assert inp['marker_oris'].reshape(B, 3, 12, 9)[0, 1, 3].abs().sum() == 0
    assert inp['marker_pos'].sum() == B * 3 * 36 - 3
    assert inp['offset_r'].shape == (B, 12, 3, 3)


def test_resnet_plumbing_on_cpu():
    """BASELINE config 0: ResNet, one 32-frame 12-sensor window, forward on PyTorch CPU."""
    cfg = Configuration.defaults(m_type='resnet', m_hidden_size=64, m_num_layers=3, use_marker_pos=True,
                                 use_marker_ori=True, n_markers=12, window_size=32)
    net = create_model(cfg, None).eval()
    b = RealBatch([0], torch.tensor([32]), torch.zeros(1, 32, 66), torch.zeros(1, 10), torch.zeros(1, 32, 3),
                  torch.randn(1, 32, 36), torch.randn(1, 32, 108), torch.ones(1, 32, 12))
    out = net(b)
    assert out['pose_hat'].shape == (1, 32, 63) and out['root_ori_hat'].shape == (1, 32, 3)


def test_evaluate_real_cli_runs_the_resnet_baseline_on_cpu():
    """BASELINE config 0 through the CLI: `scripts/evaluate_real.py` runs the frame-wise ResNet on CPU tensors without a
    GPU (plumbing only, as reference scripts/evaluate_real.py:24-61 does on `C.DEVICE` = cpu); the LGD models refuse."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, 'scripts', 'evaluate_real.py'), '--synthetic', '--device', 'cpu']
    ok = subprocess.run(cmd + ['--m_type', 'resnet', '--n_markers', '12', '--max_sequences', '1', '--json'],
                        capture_output=True, text=True, timeout=600)
    assert ok.returncode == 0, ok.stderr[-2000:]
    res = json.loads(ok.stdout.strip().splitlines()[-1])
    assert res['frames'] == 3460 and res['metrics']['MPJAE [deg]'] > 0.0
    # no body model on the CPU plumbing path: the position metrics are NaN (not a perfect-looking 0 mm)
    assert res['metrics']['MPJPE [mm]'] != res['metrics']['MPJPE [mm]']
    bad = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0 and 'need an MI355X' in (bad.stderr + bad.stdout)


def test_preprocess_refuses_unimplemented_noise_augmentation():
    from em_pose_amd.data.transforms import get_end_to_end_preprocess_fn
    from em_pose_amd.helpers.configuration import lgd_config
    cfg = lgd_config(12, True, 2, suppression_noise_length=0.1)
    with pytest.raises(NotImplementedError):
        get_end_to_end_preprocess_fn(cfg, None, [], randomize_if_configured=True)


def test_offsets_npz_file_format(tmp_path):
    """`*_offsets.npz` as the reference stores them (data/transforms.py:145-155) load into the transform's offset sets."""
    from em_pose_amd.data.transforms import load_offsets_npz
    rng = np.random.default_rng(0)
    path = str(tmp_path / 'subject_offsets.npz')
    means, covs, r = rng.normal(size=(12, 3)), rng.normal(size=(12, 3, 3)), rng.normal(size=(12, 3, 3))
    vids = np.arange(12) * 7
    np.savez(path, means=means, covs=covs, r=r, vertex_ids=vids)
    o = load_offsets_npz(path)
    np.testing.assert_array_equal(o['means'], means)
    np.testing.assert_array_equal(o['r'], r)
    np.testing.assert_array_equal(o['covs'], covs)
    assert o['vertex_ids'].tolist() == vids.tolist()


def test_extract_window_modes():
    """reference transforms.py:66-96: beginning / middle / random windows, short samples returned whole."""
    from em_pose_amd.data.data import RealSample
    from em_pose_amd.data.transforms import ExtractWindow
    f = 50
    s = RealSample('rec', np.arange(f * 36, dtype=np.float32).reshape(f, 12, 3), np.zeros((f, 12, 3, 3), np.float32),
                   np.ones((f, 12), np.float32), np.zeros((f, 66), np.float32), np.zeros(10, np.float32),
                   np.zeros((f, 3), np.float32), {'means': np.zeros((12, 3)), 'covs': np.zeros((12, 3, 3)),
                                                  'r': np.zeros((12, 3, 3))})
    first = lambda x: int(x.marker_pos_real[0, 0]) // 36
    assert first(ExtractWindow(16, mode='beginning')(s)) == 0
    mid = ExtractWindow(16, mode='middle')(s)
    assert first(mid) == 25 - 8 and mid.n_frames == 16 and mid.smpl_poses.shape == (16, 66)
    rng = np.random.RandomState(3)
    want = np.random.RandomState(3).randint(0, f - 16 + 1)
    assert first(ExtractWindow(16, rng=rng, mode='random')(s)) == want
    assert ExtractWindow(64, mode='beginning')(s) is s
    with pytest.raises(ValueError):
        ExtractWindow(16, mode='end')
    with pytest.raises(ValueError):
        ExtractWindow(16, mode='random')


def test_reference_named_helper_modules():
    """`empose.nn.loss` / `empose.helpers.utils` names resolve; local_to_global round-trips through both formats."""
    from em_pose_amd.eval.metrics import local_to_global_rotations
    from em_pose_amd.helpers.configuration import CONSTANTS as C
    from em_pose_amd.helpers.utils import count_parameters, global_oris_from_pose, local_to_global, mask_from_seq_lengths
    from em_pose_amd.nn.loss import normal_mse, padded_loss, reconstruction_loss  # noqa: F401
    rng = np.random.default_rng(0)
    poses = torch.from_numpy(rng.normal(0, 0.6, size=(7, 66)).astype(np.float32))
    rot = local_to_global(poses, C.SMPL_PARENTS, output_format='rotmat')
    want = local_to_global_rotations(poses.numpy().astype(np.float64), C.SMPL_PARENTS)
    np.testing.assert_allclose(rot.numpy().reshape(7, 22, 3, 3), want, atol=1e-6)
    aa = local_to_global(poses, C.SMPL_PARENTS)                       # global rotations as axis-angle ...
    back = local_to_global(aa, [-1] * 22, output_format='rotmat')     # ... exponentiated again without a chain
    np.testing.assert_allclose(back.numpy(), rot.numpy(), atol=1e-5)
    same = local_to_global(rot, [-1] * 22, output_format='rotmat', input_format='rotmat')
    np.testing.assert_allclose(same.numpy(), rot.numpy(), atol=0)
    oris = global_oris_from_pose(poses[:, :3].reshape(1, 7, 3), poses[:, 3:].reshape(1, 7, 63), C.SMPL_PARENTS, [0, 5])
    np.testing.assert_allclose(oris.numpy().reshape(7, 2, 9), rot.numpy().reshape(7, 22, 9)[:, [0, 5]], atol=0)
    assert mask_from_seq_lengths(torch.tensor([2, 3])).tolist() == [[True, True, False], [True, True, True]]
    assert count_parameters(torch.nn.Linear(3, 2)) == 8


def test_so3_maps_match_scipy_and_the_reference_clamp():
    from scipy.spatial.transform import Rotation
    from em_pose_amd.helpers.so3 import so3_exponential_map, so3_log_map, so3_relative_angle
    rng = np.random.default_rng(1)
    v = rng.normal(0, 1.0, size=(50, 3))
    v = v / np.linalg.norm(v, axis=1, keepdims=True) * rng.uniform(0.05, 3.0, size=(50, 1))
    R = so3_exponential_map(torch.from_numpy(v))
    np.testing.assert_allclose(R.numpy(), Rotation.from_rotvec(v).as_matrix(), atol=1e-12)
    np.testing.assert_allclose(so3_log_map(R).numpy(), v, atol=1e-9)
    np.testing.assert_allclose(so3_relative_angle(R, R).numpy(), 0.0, atol=1e-6)
    # below sqrt(eps) = 0.01 rad the squared angle is clamped (reference so3.py:115-116): the map is I + K + K^2/2 scaled
    tiny = torch.tensor([[1e-3, 0.0, 0.0]], dtype=torch.float64)
    Rt = so3_exponential_map(tiny)[0]
    assert float(Rt[2, 1]) == pytest.approx(1e-3 * np.sin(0.01) / 0.01, rel=1e-12)
    with pytest.raises(ValueError):
        so3_exponential_map(torch.zeros(3))


def test_amass_sample_and_batch(tmp_path):
    """reference data.py:311-459: npz sample, window extraction, padded collation, the input dict."""
    from em_pose_amd.data.data import AMASSBatch, AMASSSample
    from em_pose_amd.data.transforms import ExtractWindow, ToTensor
    rng = np.random.default_rng(2)
    path = str(tmp_path / 'seq.npz')
    np.savez(path, poses=rng.normal(size=(40, 156)), betas=rng.normal(size=16), trans=rng.normal(size=(40, 3)),
             mocap_framerate=np.array(120.0))
    a = AMASSSample.from_disk(path, 'a')
    assert a.poses.shape == (40, 66) and a.shape.shape == (10,) and a.fps == 120.0 and a.n_frames == 40
    b = ExtractWindow(16, mode='middle')(a)
    assert b.n_frames == 16 and np.array_equal(b.poses, a.poses[12:28])
    samples = [ToTensor()(a), ToTensor()(b)]
    batch = AMASSBatch.from_sample_list(samples)
    assert batch.poses.shape == (2, 40, 66) and batch.seq_lengths.tolist() == [40, 16] and batch.joints_gt is None
    assert float(batch.poses[1, 16:].abs().max()) == 0.0 and batch.genders == ['unknown', 'unknown']
    batch.marker_pos_synth, batch.marker_ori_synth = torch.zeros(2, 40, 36), torch.ones(2, 40, 108)
    batch.marker_pos_noisy = torch.full((2, 40, 36), 2.0)
    inp = batch.get_inputs(sf=4, ef=10)
    assert inp['marker_pos'].shape == (2, 6, 36) and float(inp['marker_pos'].min()) == 2.0   # noisy wins over synth
    assert float(inp['marker_oris'].min()) == 1.0 and inp['marker_masks'] is None and inp['joints'] is None


def _lmdb_fixture():
    z = np.load(os.path.join(H.GOLDEN, 'lmdb_schema.npz'))
    keys = json.loads(str(z['keys']))
    return z, {k.encode(): z['rec/{}'.format(j)].tobytes() for j, k in enumerate(keys)}


def test_lmdb_key_schema_reader_equals_the_reference_reader():
    """`LMDBDataset` over a dict store returns, for every sequence, exactly what the reference's reader
    (empose/data/datasets.py:42-59, run by make_golden.py) returned for the same records; the key set is the one the
    reference's conversion script writes (scripts/preprocess_amass_3dpw.py:171-189)."""
    from em_pose_amd.data.datasets import LMDBDataset, encode_sequence_records
    z, records = _lmdb_fixture()
    n = int(z['n'])
    assert set(records) == {'{}{}'.format(f, i).encode() for i in range(n) for f in
                            ('poses', 'betas', 'trans', 'joints', 'n_frames', 'id', 'gender')} | {b'__len__'}
    ds = LMDBDataset(records)          # a dict has get(key) -> bytes | None
    assert len(ds) == n
    again = {b'__len__': records[b'__len__']}
    for i in range(n):
        s, meta = ds[i], json.loads(str(z['seq{}/meta'.format(i)]))
        assert (s.id, s.gender, s.fps, s.n_frames) == (meta['id'], meta['gender'], meta['fps'], meta['n_frames'])
        for name, got in (('poses', s.poses), ('shape', s.shape), ('trans', s.trans), ('joints', s.joints)):
            want = z['seq{}/{}'.format(i, name)]
            assert got.dtype == np.float32 and got.shape == want.shape and np.array_equal(got, want), (i, name)
            assert got.flags.writeable     # the reference copies out of the read-only transaction buffer
        assert s.joints.shape == (s.n_frames, 66) and s.trans.shape == (s.n_frames, 3)
        # the writer side: decoded arrays encode back to the very same bytes
        back = np.frombuffer(records['joints{}'.format(i).encode()], dtype=np.float32).reshape(s.n_frames, -1)
        again.update(encode_sequence_records(i, s.id, s.poses, s.shape, s.trans, back, s.gender))
    assert again == records
    # the sample feeds the training-side containers like an npz sample does
    from em_pose_amd.data.data import AMASSBatch
    samples = [ds[i].extract_window(0, 1) for i in (0, 2)]
    for smp in samples:
        smp.poses = smp.poses[:, :C.MAX_INDEX_ROOT_AND_BODY]
        smp.shape = smp.shape[:C.N_SHAPE_PARAMS]
        smp.to_tensor()
    batch = AMASSBatch.from_sample_list(samples)
    assert batch.poses.shape == (2, 1, 66) and batch.joints_gt.shape == (2, 1, 66)


def test_lmdb_reader_errors_and_transform():
    from em_pose_amd.data.datasets import LMDBDataset
    _, records = _lmdb_fixture()
    with pytest.raises(KeyError):
        LMDBDataset({})
    broken = dict(records)
    del broken[b'trans1']
    ds = LMDBDataset(broken)
    with pytest.raises(KeyError, match='trans1'):
        ds[1]
    broken = dict(records)
    broken[b'n_frames0'] = b'5'       # 7 frames of data
    with pytest.raises(ValueError):
        LMDBDataset(broken)[0]
    with pytest.raises(IndexError):
        ds[len(ds)]
    assert LMDBDataset(records, transform=lambda s: s.n_frames)[2] == 33
    with pytest.raises(ImportError, match='lmdb'):   # a path needs the real package, which this image lacks
        LMDBDataset('/nonexistent/amass_lmdb')


def test_handle_key_tensor_list_is_cached_and_follows_registrations():
    """The LGD model keys its packed device handle on its parameter / buffer tensors; the list is cached (walking the
    module tree per forward cost more than a streaming chunk's host side) and must be rebuilt when a tensor object is
    replaced -- seen through the process-wide registration hooks -- or tensors move (`_apply`, `load_state_dict`)."""
    net = create_model(lgd_config(12, True, 2, hidden=32, rnn_hidden=32), SMPLLayer(H.small_model()))
    first = net._own_parameters()
    assert net._own_parameters() is first                       # cached
    assert not any(t is p for p in net.smpl.parameters() for t in first)
    assert len(first) == len([n for n, _ in net.named_parameters() if not n.startswith('smpl.')]) + \
        len([n for n, _ in net.named_buffers() if not n.startswith('smpl.')])
    lin = net.pose_net_iter.hidden_to_output
    new_w = torch.nn.Parameter(torch.zeros_like(lin.weight))
    lin.weight = new_w                                           # a replaced Parameter: new object, same name
    second = net._own_parameters()
    assert second is not first and any(t is new_w for t in second)
    net.load_state_dict(net.state_dict())
    assert net._own_parameters() is not second
    third = net._own_parameters()
    net.double()
    assert net._own_parameters() is not third


def test_metrics_engine_accepts_a_precomputed_valid_mask():
    """`MetricsEngine.compute(..., valid=...)`: a caller that has lengths and masks on the host (the streaming driver)
    hands over the frames that count; same accumulators as letting the engine derive them."""
    from em_pose_amd.eval.metrics import MetricsEngine
    rng = np.random.default_rng(3)
    n, f = 3, 7
    t = lambda *s: torch.as_tensor(rng.normal(0, 0.3, size=s), dtype=torch.float32)
    pose, pose_hat, shape = t(n, f, 63), t(n, f, 63), t(n, 10)
    root, root_hat = t(n, f, 3), t(n, f, 3)
    lens = torch.tensor([7, 3, 5])
    masks = torch.ones(n, f, 12)
    masks[0, 2, 4] = 0.0
    a, b = MetricsEngine(None), MetricsEngine(None)
    a.compute(pose, shape, pose_hat, None, lens, root, root_hat, frame_mask=masks)
    valid = MetricsEngine.valid_frames(lens, n, f, masks)
    assert int(valid.sum()) == 7 + 3 + 5 - 1
    b.compute(pose, shape, pose_hat, None, lens, root, root_hat, frame_mask=masks, valid=valid)
    sa, sb = a.state(), b.state()
    for k in sa:
        np.testing.assert_array_equal(sa[k], sb[k])
    assert sa['angle'].shape[0] == 14


def test_prelu_flip_proof_accepts_a_recorded_element_and_nothing_else():
    """tests/helpers.py::explain_by_prelu_flips (what the full-width training test accepts a BatchNorm gradient entry
    beyond tolerance by): a difference that IS a flip of a recorded near-zero element is explained -- as one element, and
    as two in different columns --, the same magnitude at a column without a recorded element, a recorded element at half
    its predicted magnitude, and a dense difference are not."""
    import os
    from tests import helpers as TH
    z = np.load(os.path.join(TH.GOLDEN, 'train_fp_lgdrnn12_n4_h512.npz'))
    fp = {k: z[k] for k in z.files if k.startswith('prelu/pose_net_iter.hidden_layers.1.layers.2/')}
    name, slope = 'pose_net_iter.hidden_layers.1.layers.1.bias', 0.3
    rng = np.random.default_rng(0)
    ref = rng.normal(size=512) * 1e-2
    want = TH.tensor_fingerprint(name, ref)
    tol_e, tol_p, tol_l = 1e-6, 1e-6 * np.sqrt(512), 1e-6
    cands = []
    for c in range(4):
        key = 'prelu/pose_net_iter.hidden_layers.1.layers.2/%d/' % c
        for col, zz, cot in zip(fp[key + 'col'], fp[key + 'z'], fp[key + 'cot']):
            if abs(zz) <= float(fp[key + 'z_noise']) and abs(cot) * (1 - slope) > 20 * tol_e:
                cands.append((int(col), -np.sign(zz) * (1 - slope) * cot))
    assert len(cands) >= 2
    noise = rng.normal(size=512) * 1e-7
    one = ref + noise
    one[cands[0][0]] += cands[0][1]
    got = TH.explain_by_prelu_flips(fp, name, one, want, tol_e, tol_p, tol_l, slope)
    assert got is not None and len(got) == 1 and got[0][2] == cands[0][0]
    other = next(c for c in cands if c[0] != cands[0][0])
    two = one.copy()
    two[other[0]] += other[1]
    got = TH.explain_by_prelu_flips(fp, name, two, want, tol_e, tol_p, tol_l, slope)
    assert got is not None and sorted(g[2] for g in got) == sorted([cands[0][0], other[0]])
    taken = {int(c) for k in fp if k.endswith('/col') for c in fp[k]}
    free = next(c for c in range(512) if c not in taken)
    wrong_col = ref + noise
    wrong_col[free] += cands[0][1]
    assert TH.explain_by_prelu_flips(fp, name, wrong_col, want, tol_e, tol_p, tol_l, slope) is None
    half = ref + noise
    half[cands[0][0]] += 0.5 * cands[0][1]
    assert TH.explain_by_prelu_flips(fp, name, half, want, tol_e, tol_p, tol_l, slope) is None
    assert TH.explain_by_prelu_flips(fp, name, ref + 30 * tol_e, want, tol_e, tol_p, tol_l, slope) is None
    # a Linear weight or a PReLU slope is never explained this way
    assert TH.explain_by_prelu_flips(fp, 'pose_net_iter.hidden_layers.1.layers.2.weight', one[:1], want, 1, 1, 1, slope) is None
