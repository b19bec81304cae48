"""CPU tests of the evaluation support code: metrics vs reference vectors, transforms, sequence sharding, and the
world_size=2 gather of metric accumulators over gloo (the N>1 path of BASELINE configs[3])."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from em_pose_amd import synthetic
from em_pose_amd.data.data import RealBatch, RealSample
from em_pose_amd.data.transforms import NormalizeRealMarkers, NormalizeRoot, ToTensor, matrix_to_rotvec
from em_pose_amd.eval.helpers import partition_sequences, window_generator
from em_pose_amd.eval.metrics import (MetricsEngine, geodesic_degrees, local_to_global_rotations, procrustes_align,
                                      rotvec_to_matrix)
from em_pose_amd.helpers.configuration import CONSTANTS as C
from tests import helpers as H


def test_metrics_match_reference_vectors():
    """MPJPE / PA-MPJPE recorded from the reference's MetricsEngine.compute_joint_dist + get_metrics."""
    z = np.load(os.path.join(H.GOLDEN, 'components.npz'))
    me = MetricsEngine(None)
    me.compute_joint_dist(torch.from_numpy(z['me_joints']), torch.from_numpy(z['me_joints_hat']),
                          torch.from_numpy(z['me_len']), torch.from_numpy(z['me_mask']))
    got = me.get_metrics()
    for k in ('MPJPE [mm]', 'MPJPE STD', 'PA-MPJPE [mm]', 'PA-MPJPE STD'):
        np.testing.assert_allclose(got[k], float(z['me_' + k.split(' ')[0] + ('_STD' if 'STD' in k else '')]),
                                   rtol=1e-6)
    assert z['me_n_rows'] == np.concatenate(me.eucl_dists).shape[0]


def test_procrustes_recovers_similarity_transform():
    rng = np.random.default_rng(0)
    X = rng.normal(size=(5, 22, 3))
    R = synthetic._exp_so3(rng.normal(size=(5, 3)))
    Y = 1.7 * X @ np.swapaxes(R, 1, 2) + rng.normal(size=(5, 1, 3))
    np.testing.assert_allclose(procrustes_align(X, Y), X, atol=1e-9)


def test_rotation_helpers():
    rng = np.random.default_rng(1)
    r = rng.normal(0, 1.0, size=(50, 3))
    R = rotvec_to_matrix(r)
    np.testing.assert_allclose(R, synthetic._exp_so3(r), atol=1e-12)
    np.testing.assert_allclose(rotvec_to_matrix(matrix_to_rotvec(R)), R, atol=1e-9)
    near_pi = rng.normal(size=(8, 3))
    near_pi = near_pi / np.linalg.norm(near_pi, axis=1, keepdims=True) * (np.pi - 1e-6)
    Rp = rotvec_to_matrix(near_pi)
    np.testing.assert_allclose(rotvec_to_matrix(matrix_to_rotvec(Rp)), Rp, atol=1e-6)
    assert abs(geodesic_degrees(np.eye(3), rotvec_to_matrix(np.array([0.0, 0.3, 0.0]))) - np.rad2deg(0.3)) < 1e-9
    g = local_to_global_rotations(rng.normal(0, 0.3, size=(4, 66)), C.SMPL_PARENTS)
    np.testing.assert_allclose(g @ np.swapaxes(g, -1, -2), np.broadcast_to(np.eye(3), g.shape), atol=1e-12)


def _sample(n_frames, seed):
    rng = np.random.default_rng(seed)
    return RealSample('s%d' % seed, rng.normal(size=(n_frames, 12, 3)).astype(np.float32),
                      synthetic._exp_so3(rng.normal(size=(n_frames, 12, 3))).astype(np.float32),
                      np.ones((n_frames, 12), np.float32), rng.normal(0, 0.3, size=(n_frames, 66)).astype(np.float32),
                      rng.normal(size=16).astype(np.float32), rng.normal(size=(n_frames, 3)).astype(np.float32),
                      {'means': np.zeros((12, 3), np.float32), 'covs': np.zeros((12, 3, 3), np.float32),
                       'r': np.tile(np.eye(3, dtype=np.float32), (12, 1, 1))})


def test_real_sample_pipeline_and_chunking(tmp_path):
    s = _sample(600, 3)
    raw_pos, raw_ori = s.marker_pos_real.copy(), s.marker_ori_real.copy()
    R0 = rotvec_to_matrix(s.smpl_poses[0, :3].astype(np.float64))
    b = NormalizeRoot()(RealBatch.from_sample_list([ToTensor()(NormalizeRealMarkers()(s))]))
    # sensors are expressed in the frame of the first root pose, translation removed
    want = (raw_pos.reshape(600, 12, 3) - s.smpl_trans.numpy()[:, None]) @ R0
    np.testing.assert_allclose(b.marker_pos_real[0].numpy().reshape(600, 12, 3), want, atol=1e-5)
    want_o = R0.T @ raw_ori.reshape(600, 12, 3, 3)
    np.testing.assert_allclose(b.marker_ori_real[0].numpy().reshape(600, 12, 3, 3), want_o, atol=1e-5)
    # the first root orientation is the identity after NormalizeRoot, translation is zero
    np.testing.assert_allclose(b.poses[0, 0, :3].numpy(), 0.0, atol=1e-6)
    assert float(b.trans.abs().sum()) == 0.0 and b.shapes.shape == (1, 10)
    chunks = list(window_generator(b, 256))
    assert [c.seq_length for c in chunks] == [256, 256, 88] and [int(c.seq_lengths[0]) for c in chunks] == [256, 256, 88]
    assert chunks[2].offset_r.shape == (1, 12, 3, 3)
    # *_clean.npz round trip
    d = synthetic.make_sequence(40, 1, lambda p, be, r, t: (np.zeros((40, 12, 3)), np.tile(np.eye(3), (40, 12, 1, 1))))
    path = str(tmp_path / 'x_clean.npz')
    np.savez(path, **d)
    s2 = RealSample.from_npz_clean(path)
    assert s2.n_frames == 40 and s2.marker_ori_real.shape == (40, 108)


def test_partition_sequences_lpt():
    lengths = synthetic.README_SEQUENCE_LENGTHS
    assert len(lengths) == 36 and sum(lengths) == 54030  # reference README.md:107-142
    parts = partition_sequences(lengths, 8)
    assert sorted(i for p in parts for i in p) == list(range(36))
    loads = [sum(lengths[i] for i in p) for p in parts]
    assert max(loads) <= 1.08 * (54030 / 8) and max(loads) >= max(lengths)
    assert partition_sequences(lengths, 1) == [list(range(36))]
    assert partition_sequences([5, 5, 5], 2) == [[0, 2], [1]]


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, lengths, q):
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        me = MetricsEngine(None)
        for i in partition_sequences(lengths, world)[rank]:
            rng = np.random.default_rng(i)
            j = torch.from_numpy(rng.normal(size=(1, lengths[i], 66)))
            me.compute_joint_dist(j, j + torch.from_numpy(rng.normal(0, 0.02, size=(1, lengths[i], 66))))
        me.gather()
        q.put((rank, me.get_metrics(), np.concatenate(me.eucl_dists).shape[0]))
    finally:
        dist.destroy_process_group()


def test_metric_gather_world_size_2_equals_single_process():
    lengths = [7, 3, 5, 2, 9]
    single = MetricsEngine(None)
    order = [i for p in partition_sequences(lengths, 2) for i in p]  # rank-major order of the gathered rows
    for i in order:
        rng = np.random.default_rng(i)
        j = torch.from_numpy(rng.normal(size=(1, lengths[i], 66)))
        single.compute_joint_dist(j, j + torch.from_numpy(rng.normal(0, 0.02, size=(1, lengths[i], 66))))
    want = single.get_metrics()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, lengths, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, got, n_rows in results:
        assert n_rows == sum(lengths)
        for k in want:
            assert got[k] == pytest.approx(want[k], rel=1e-12, abs=1e-12, nan_ok=True), (rank, k)


def _grad_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    os.environ['RANK'], os.environ['WORLD_SIZE'] = str(rank), str(world)
    from em_pose_amd.helpers.distributed import allreduce_gradients, init_from_env
    init_from_env(torch.device('cpu'))
    try:
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(40, 300), torch.nn.Linear(300, 7))
        frozen = torch.nn.Parameter(torch.zeros(3), requires_grad=False)
        x = torch.full((4, 40), float(rank + 1))
        net(x).sum().backward()
        net[1].bias.grad = None if rank == 1 else net[1].bias.grad  # a parameter without gradient on one rank
        n_buckets = allreduce_gradients(list(net.parameters()) + [frozen], bucket_bytes=16 << 10)
        first = [p.grad.numpy().copy() for p in net.parameters()]
        # the persistent form: buckets kept across steps, parameters staged one by one as a reverse sweep would finish
        # them (last layer first), `.grad` is the bucket slice afterwards and stays at the same address
        from em_pose_amd.helpers.distributed import GradientBuckets
        order = list(net.parameters())[::-1]
        buckets = GradientBuckets(order, bucket_bytes=4 << 10)
        ptrs, second = None, None
        for step in range(2):
            for p in net.parameters():
                p.grad = None
            net(x).sum().backward()
            if rank == 1:
                net[1].bias.grad = None
            for p in order:
                buckets.stage(p)
            assert buckets.finish() == buckets.n_buckets >= 2
            now = [p.grad.data_ptr() for p in net.parameters()]
            assert all(p.grad.data_ptr() == buckets.view_of(p).data_ptr() for p in net.parameters())
            assert ptrs is None or ptrs == now
            ptrs, second = now, [p.grad.numpy().copy() for p in net.parameters()]
        assert all(np.array_equal(a, b) for a, b in zip(first, second))
        q.put((rank, n_buckets, first))  # plain arrays: no fd passing
    finally:
        dist.destroy_process_group()


def test_gradient_allreduce_world_size_2():
    """Bucketed gradient averaging over gloo equals the mean of the per-rank gradients."""
    from em_pose_amd.helpers.distributed import shard_range
    assert [shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(40, 300), torch.nn.Linear(300, 7))
    want = []
    for rank in range(2):
        net.zero_grad()
        net(torch.full((4, 40), float(rank + 1))).sum().backward()
        g = [p.grad.clone() for p in net.parameters()]
        if rank == 1:
            g[3] = torch.zeros_like(g[3])
        want.append(g)
    mean = [(a + b) / 2 for a, b in zip(*want)]
    assert res[0][1] == res[1][1] >= 2  # several buckets, same number on every rank
    for rank, _, grads in res:
        for got, w in zip(grads, mean):
            np.testing.assert_allclose(got, w.numpy(), rtol=1e-6, atol=1e-6)
