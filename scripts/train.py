#!/usr/bin/env python
"""
Training driver for the LGD models (mirror of the loop in reference scripts/train.py:125-152: zero_grad -> forward ->
model.backward -> optimizer.step, Adam, per-step loss line) on SYNTHETIC windows -- AMASS / 3DPW are not available here
(SURVEY.md 8c), so ground-truth poses are random-walk windows and the sensor readings come from the HIP body model
(the role of SMPLFK + SampleMarkersWithOffsets in the reference).  BASELINE.json configs[4].

    python scripts/train.py --steps 20                       # LGD-RNN-12, N=4, ws=32, 12 windows per GPU
    python scripts/train.py --steps 20 --gpus 8              # spawns 8 ranks; torch.distributed.run works too

Data parallel over windows; gradients averaged with bucketed RCCL all-reduces (em_pose_amd/helpers/distributed.py).
Forward, losses, backward and the optimiser step are the library's own kernels (em_pose_amd/nn/train_engine.py,
em_pose_amd/helpers/optim.py): no autograd graph, no library GEMM / RNN / elementwise kernel on the step.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import numpy as np  # noqa: E402
import torch  # noqa: E402

from em_pose_amd import synthetic  # noqa: E402
from em_pose_amd.bodymodels.smpl import SMPLLayer  # noqa: E402
from em_pose_amd.data.data import SyntheticBatch  # noqa: E402
from em_pose_amd.helpers.configuration import lgd_config  # noqa: E402
from em_pose_amd.helpers.distributed import (GradientBuckets, allreduce_gradients, attach_gradient_buckets,  # noqa: E402
                                             init_from_env)
from em_pose_amd.nn.models import create_model  # noqa: E402


def _body_columns(sample):
    """LMDB records keep all 156 AMASS pose columns and 16 shape coefficients; the batch takes root + body, 10 betas."""
    from em_pose_amd.helpers.configuration import CONSTANTS as C
    sample.poses, sample.shape = sample.poses[:, :C.MAX_INDEX_ROOT_AND_BODY], sample.shape[:C.N_SHAPE_PARAMS]
    return sample


def make_buckets(net, params, world, args):
    """Persistent flat gradient buckets in the order the reverse sweep finishes them (helpers/distributed.py); with more
    than one rank -- or `--force_dist`, the single-rank self-test of the RCCL path -- the engine writes gradients straight
    into them and each bucket's all-reduce overlaps the rest of the sweep."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or not (world > 1 or args.force_dist):
        return None
    from em_pose_amd.nn.train_engine import LgdTrainEngine
    order = LgdTrainEngine.gradient_order(net) if LgdTrainEngine.supported(net) else []
    rest = [p for p in params if id(p) not in {id(q) for q in order}]
    buckets = GradientBuckets(order + rest, bucket_bytes=args.bucket_mb << 20, force=args.force_dist)
    attach_gradient_buckets(net, buckets)
    return buckets


def average_gradients(buckets, params):
    return buckets.finish() if buckets is not None else allreduce_gradients(params)


def train_on_amass(args, dev, rank, world):
    """Data-parallel training on AMASS npz sequences: random windows, offsets with the configured noise level, periodic
    validation on held-out sequences, best checkpoint + config.json in the reference's experiment layout (so that
    `scripts/evaluate_real.py --model_id <id>` / `eval.helpers.load_model` find them)."""
    import glob
    from torch.utils.data import DataLoader, Subset
    from em_pose_amd import _lib
    from em_pose_amd.data.data import AMASSBatch
    from em_pose_amd.data.datasets import AMASSNpzDataset, LMDBDataset
    from em_pose_amd.data.transforms import ExtractWindow, ToTensor, get_end_to_end_preprocess_fn
    from em_pose_amd.eval.helpers import evaluate
    from em_pose_amd.eval.metrics import MetricsEngine
    from em_pose_amd.helpers.configuration import CONSTANTS as C
    from em_pose_amd.helpers.utils import count_parameters, create_model_dir

    smpl = SMPLLayer(args.smpl_model if args.smpl_model else synthetic.make_model()).to(dev)
    torch.manual_seed(args.seed)
    cfg = lgd_config(args.n_markers, not args.no_rnn, args.iterations, window_size=args.window_size, lr=args.lr,
                     offset_noise_level=args.offset_noise_level)
    net = create_model(cfg, smpl).to(dev)
    params = [q for n, q in net.named_parameters() if not n.startswith('smpl.')]
    opt = torch.optim.Adam(params, lr=args.lr)
    offsets = args.offset_files or sorted(glob.glob(os.path.join(C.DATA_DIR_TEST or '.', '*_offsets.npz')))
    if not offsets:
        raise SystemExit('no *_offsets.npz files: pass --offset_files or set EM_DATA_REAL')
    buckets = make_buckets(net, params, world, args)
    fn_train = get_end_to_end_preprocess_fn(cfg, smpl, offsets, randomize_if_configured=True)
    fn_valid = get_end_to_end_preprocess_fn(cfg, smpl, offsets, randomize_if_configured=False)

    # The sequences: AMASS npz files under --amass_dir, or the records of an LMDB database in the reference's key schema
    # (--amass_lmdb, + optionally a separate --valid_lmdb as the reference trains on AMASS and validates on 3DPW).
    def ex_seqs(transform, items, lmdb_path=None):
        if lmdb_path is None:
            return AMASSNpzDataset(None, transform, files=items)
        return Subset(LMDBDataset(lmdb_path, transform), items)
    valid_src = train_src = args.amass_lmdb
    if args.amass_lmdb:
        items = list(range(len(LMDBDataset(args.amass_lmdb))))
    else:
        items = sorted(glob.glob(os.path.join(args.amass_dir, '**', '*.npz'), recursive=True))
    if args.valid_lmdb:
        valid_src, valid_items = args.valid_lmdb, list(range(len(LMDBDataset(args.valid_lmdb))))
    else:
        if len(items) < 2:
            raise SystemExit('need at least two AMASS sequences')
        n_valid = max(1, int(round(len(items) * args.valid_fraction)))
        valid_items = items[::max(1, len(items) // n_valid)][:n_valid]
        held_out = set(valid_items)
        items = [f for f in items if f not in held_out]
    # Sequences sharded over the ranks.  Every rank must run the same number of steps per epoch (each step ends in a
    # gradient all-reduce): the list is cut to a multiple of world x batch size before it is dealt out.
    usable = (len(items) // (world * args.bs_train)) * world * args.bs_train
    if usable == 0:
        raise SystemExit('{} training sequences are fewer than one batch of {} on each of {} ranks'
                         .format(len(items), args.bs_train, world))
    train_items = items[:usable][rank::world]
    win = lambda mode, rng=None: (lambda smp: ToTensor()(ExtractWindow(args.window_size, rng=rng, mode=mode)(
        _body_columns(smp))))
    window_rng = np.random.RandomState(args.seed + rank)

    def seed_worker(worker_id):
        # Every DataLoader worker owns a copy of window_rng.  Workers are re-created every epoch; torch.initial_seed()
        # inside a worker is that epoch's base seed + worker_id, so the crop offsets differ between epochs, workers and
        # (through the rank term: the base seed is the same on every rank) ranks.
        window_rng.seed((torch.initial_seed() + 1000003 * (rank + 1)) % (2 ** 32))
    train_data = ex_seqs(win('random', window_rng), train_items, train_src)
    valid_data = ex_seqs(win('middle'), valid_items, valid_src)
    loader = lambda data, bs, shuffle: DataLoader(data, batch_size=bs, shuffle=shuffle, num_workers=args.data_workers,
                                                 collate_fn=AMASSBatch.from_sample_list, drop_last=shuffle,
                                                 worker_init_fn=seed_worker if shuffle else None)
    train_loader, valid_loader = loader(train_data, args.bs_train, True), loader(valid_data, args.bs_train, False)

    model_dir = checkpoint = None
    if rank == 0:
        exp_dir = args.experiment_dir or C.EXPERIMENT_DIR
        if not exp_dir:
            raise SystemExit('pass --experiment_dir or set EM_EXPERIMENTS')
        exp_id = args.experiment_id if args.experiment_id is not None else int(time.time())
        model_dir = create_model_dir(exp_dir, exp_id, net.model_name())
        cfg.to_json(os.path.join(model_dir, 'config.json'))
        checkpoint = os.path.join(model_dir, 'model.pth')
        print('Model created with {} trainable parameters'.format(count_parameters(net)))
        print('Saving checkpoints to {}'.format(checkpoint))
    me = MetricsEngine(smpl)
    step, best, done, skipped = 0, float('inf'), False, 0
    for epoch in range(args.n_epochs):
        for i, abatch in enumerate(train_loader):
            t0 = time.perf_counter()
            net.train()
            opt.zero_grad()
            batch = fn_train(abatch.to_gpu(dev))
            try:
                loss, vals = net.backward(batch, net(batch))
                poisoned = _lib.lib().empose_async_status() != 0
            except _lib.EmposeError:
                # (an entry point of the same step already saw the word: EMPOSE_ETIMEOUT; anything else is not ours to hide)
                if _lib.lib().empose_async_status() == 0:
                    raise
                poisoned = True
            # `vals` are host floats: the step's kernels have finished, so the word the cooperative kernels (whole-sequence
            # LSTM, one-launch training layers) count their abandoned polls in is final and reading it costs nothing.  A
            # step whose poll gave up carries NaN gradients: it is NOT applied.  Every rank decides the same way (the
            # collective below must be entered by all or by none), so the flags are reduced first.
            if world > 1:
                flag = torch.tensor([float(poisoned)], device=dev)
                torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
                poisoned = bool(flag.item())
            if poisoned:
                skipped += 1
                print('[TRAIN {:0>5d} | {:0>3d}] rank {}: step dropped, a cooperative kernel gave up on a poll ({}); '
                      'gradients discarded'.format(i + 1, epoch + 1, rank, _lib.lib().empose_last_error().decode()))
                if skipped >= 3:
                    raise SystemExit('three steps dropped: is another process using this GPU?  --option train_cols=0 '
                                     '--option lstm_persist=0 run without cooperative kernels')
                if buckets is not None:
                    buckets.finish()       # the collectives the backward already started are drained; their sums are dropped
                continue
            average_gradients(buckets, params)
            opt.step()
            if rank == 0:
                print('[TRAIN {:0>5d} | {:0>3d}] '.format(i + 1, epoch + 1) +
                      ' '.join('{}: {:.6f}'.format(k, v) for k, v in vals.items()) +
                      ' elapsed: {:.3f} secs'.format(time.perf_counter() - t0))
            step += 1
            if step % args.eval_every == 0 or step == args.steps:
                losses = evaluate(valid_loader, net, fn_valid, me, device=dev)     # every rank: same held-out set
                if rank == 0:
                    better = losses['total_loss'] < best
                    print('[VALID {:0>5d} | {:0>3d}] '.format(i + 1, epoch + 1) +
                          ' '.join('{}: {:.6f}'.format(k, v) for k, v in losses.items()) + (' ***' if better else ''))
                    print(me.to_pretty_string(me.get_metrics(), 'VALID'))
                    if better:
                        best = losses['total_loss']
                        torch.save({'iteration': i, 'epoch': epoch, 'global_step': step,
                                    'model_state_dict': net.state_dict(), 'optimizer_state_dict': opt.state_dict(),
                                    'valid_loss': best}, checkpoint)
            if args.steps and step >= args.steps:
                done = True
                break
        if done:
            break
    import torch.distributed as dist
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and args.json:
        print(json.dumps({'steps': step, 'best_valid_loss': best, 'model_dir': model_dir}))


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--steps', type=int, default=20)
    p.add_argument('--warmup', type=int, default=2)
    p.add_argument('--bs_train', type=int, default=12, help='windows per GPU (reference README.md:221: 12)')
    p.add_argument('--window_size', type=int, default=32)
    p.add_argument('--n_markers', type=int, default=12)
    p.add_argument('--iterations', type=int, default=4)
    p.add_argument('--no_rnn', action='store_true')
    p.add_argument('--lr', type=float, default=0.0005)
    p.add_argument('--seed', type=int, default=1615200973)
    p.add_argument('--graph', action='store_true',
                   help='capture forward + backward in a HIP graph and replay it (static shapes, full-length windows): the '
                        'default of the synthetic step benchmark on one stream')
    p.add_argument('--no_graph', action='store_true', help='synthetic step benchmark: eager launches (the step is then bound by '
                                                          'the host: ~250 launches at 12 windows)')
    p.add_argument('--json', action='store_true')
    # training on AMASS sequences (the loop of reference scripts/train.py:125-230 with checkpoints and validation)
    p.add_argument('--amass_dir', default=None, help='directory tree of AMASS *.npz sequences; switches from the '
                                                     'synthetic step benchmark to real training')
    p.add_argument('--force_dist', action='store_true', help='initialise the process group (RCCL) and run the gradient '
                                                           'collectives even with one rank (self-test)')
    p.add_argument('--gpus', type=int, default=1, help='data-parallel over this many GPUs of the node (spawns one rank '
                                                      'per GPU)')
    p.add_argument('--single_stream', action='store_true',
                   help='A/B: the training engine on one stream (default: shape network, weight gradients on a side stream)')
    p.add_argument('--streams', default=None, help="A/B: which parts of the step use the engine's side stream, e.g. 'bwd,wgrad' "
                                                   "(default: fwd,bwd,bwd3,wgrad)")
    p.add_argument('--weight_t', action='store_true',
                   help='A/B: transposed weight copies every step even where the reverse sweep reads W itself (<= 512 rows)')
    p.add_argument('--side_min_frames', type=int, default=None,
                   help="A/B: frames per step from which the engine's side streams are used (default 2048; 0 = always)")
    p.add_argument('--bucket_mb', type=int, default=8, help='size of a flat gradient bucket')
    p.add_argument('--option', action='append', default=[], metavar='NAME=INT',
                   help='kernel-variant switch of the library (empose_set_option), e.g. train_fused=0; repeatable')
    p.add_argument('--amass_lmdb', default=None, help='LMDB database in the reference key schema (needs `lmdb`)')
    p.add_argument('--valid_lmdb', default=None, help='validation LMDB (e.g. 3DPW); default: held-out training sequences')
    p.add_argument('--offset_files', nargs='*', default=None, help='*_offsets.npz files (default: $EM_DATA_REAL/*_offsets.npz)')
    p.add_argument('--smpl_model', default=None, help='SMPL-H model.npz (default: the synthetic stand-in body model)')
    p.add_argument('--experiment_dir', default=None, help='where <id>-<name>/model.pth, config.json go ($EM_EXPERIMENTS)')
    p.add_argument('--experiment_id', type=int, default=None)
    p.add_argument('--n_epochs', type=int, default=1)
    p.add_argument('--eval_every', type=int, default=200)
    p.add_argument('--valid_fraction', type=float, default=0.1)
    p.add_argument('--offset_noise_level', type=int, default=0)
    p.add_argument('--data_workers', type=int, default=0)
    args = p.parse_args()
    # `--gpus N` as a plain command spawns its own N ranks (helpers/distributed.py); under torch.distributed.run the ranks
    # exist already.  --force_dist sends one rank through the same spawn + RCCL path (single-GPU self-test).
    from em_pose_amd.helpers import distributed as D
    D.maybe_self_launch(os.path.abspath(__file__), args.gpus, force=args.force_dist, what='train.py')
    launched = D.launched_by_a_launcher()
    world = int(os.environ.get('WORLD_SIZE', '1')) if launched else 1
    rank = int(os.environ.get('RANK', '0')) if launched else 0
    local_rank = int(os.environ.get('LOCAL_RANK', '0')) if launched else 0
    if args.gpus not in (1, world):
        raise SystemExit('--gpus {} but WORLD_SIZE={} (RANK / WORLD_SIZE inherited from the environment, e.g. a SLURM step or '
                         'a parent launcher\'s shell, count as "already launched": unset them or pass the matching --gpus)'
                         .format(args.gpus, world))
    joined = False
    if launched and D.dist_backend(torch.device('cuda')) == 'gloo':   # device-less rendezvous: the launcher's CPU test
        dist_, rank, world = D.init_process_group(None, log=lambda m: print(m, file=sys.stderr, flush=True))
        dist_.barrier()
        joined = True
    if not torch.cuda.is_available():
        raise SystemExit('train.py runs the HIP path and needs an MI355X; there is no CPU fallback.')
    dev = torch.device('cuda', D.local_device_index(local_rank, 'train.py'))
    torch.cuda.set_device(dev)
    if not joined and launched and (world > 1 or args.force_dist):
        _, rank, world = D.init_process_group(dev)
    for kv in args.option:
        from em_pose_amd import _lib
        name, value = kv.split('=')
        _lib.check(_lib.lib().empose_set_option(name.encode(), int(value)))

    if args.streams is not None:
        from em_pose_amd.nn.train_engine import LgdTrainEngine
        LgdTrainEngine.side_parts = tuple(x for x in args.streams.split(',') if x)
    if args.single_stream:
        from em_pose_amd.nn.train_engine import LgdTrainEngine
        LgdTrainEngine.two_streams = False
    if args.weight_t:
        from em_pose_amd.nn.train_engine import _MlpView
        _MlpView.always_transpose = True
    if args.side_min_frames is not None:
        from em_pose_amd.nn.train_engine import LgdTrainEngine
        LgdTrainEngine.two_streams_min_frames = args.side_min_frames
    if args.amass_dir or args.amass_lmdb:
        return train_on_amass(args, dev, rank, world)
    model = synthetic.make_model()
    torch.manual_seed(args.seed)  # identical initial replicas on every rank
    cfg = lgd_config(args.n_markers, not args.no_rnn, args.iterations, window_size=args.window_size, lr=args.lr)
    net = create_model(cfg, SMPLLayer(model)).to(dev)
    params = [q for n, q in net.named_parameters() if not n.startswith('smpl.')]
    from em_pose_amd.helpers.optim import HipAdam
    opt = HipAdam(params, lr=args.lr)   # torch.optim.Adam semantics, one kernel launch per step
    if rank == 0:
        print('Model created with {} trainable parameters'.format(sum(q.numel() for q in net.parameters())))

    B, F = args.bs_train, args.window_size
    net.eval()

    def sensors(poses, betas, o_r, o_t):
        pos, ori, _ = net.get_estimated_real_markers(torch.from_numpy(poses).to(dev), torch.from_numpy(betas).to(dev),
                                                     torch.from_numpy(o_r[::F].copy()).to(dev),
                                                     torch.from_numpy(o_t[::F].copy()).to(dev), frames_per_window=F)
        return pos.cpu().numpy(), ori.cpu().numpy()

    def make_batch(step):
        w = synthetic.make_windows(B, F, 7919 * step + rank, sensors)
        b = SyntheticBatch(w, device=dev)
        with torch.no_grad():
            _, _, joints = net.get_estimated_real_markers(b.poses.reshape(B * F, 66),
                                                          b.shapes[:, None].expand(B, F, 10).reshape(B * F, 10),
                                                          b.offset_r_augmented, b.offset_t_augmented,
                                                          frames_per_window=F)
        b.joints_gt = joints.reshape(B, F, 66)
        return b
    batches = [make_batch(s) for s in range(4)]  # data preparation is not part of the measured step
    net.train()
    buckets = make_buckets(net, params, world, args)
    # Fixed shapes, one rank, one stream: the step replays as a HIP graph unless asked otherwise (the
    # eager step at the reference's 12 windows is bound by the host's ~250 launches, 3.1 ms against 2.4 ms of kernels).
    if not args.graph and not args.no_graph and world == 1 and not args.force_dist and args.streams is None \
            and args.side_min_frames is None and B * F < 2048:
        args.graph = True
    graphed = None
    if args.graph:
        from em_pose_amd.helpers.graphed import GraphedTrainStep
        graphed = GraphedTrainStep(net, opt, batches[0])
    times, n_coll, enq = [], 0, []
    for step in range(args.warmup + args.steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        batch = batches[step % len(batches)]
        if graphed is not None:
            vals = graphed(batch)            # forward + backward replayed; gradients in the static .grad tensors
            n_coll = average_gradients(buckets, params)
            opt.step()
            torch.cuda.synchronize()
            vals = {k: float(v) for k, v in vals.items()}
        else:
            opt.zero_grad()
            out = net(batch)
            loss, vals = net.backward(batch, out)     # finished buckets are already being averaged on a side stream
            n_coll = average_gradients(buckets, params)
            opt.step()
            enq.append(time.perf_counter() - t0)     # host time to enqueue the whole step (before waiting for the device)
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        from em_pose_amd import _lib as _L
        _L.check(_L.lib().empose_async_status())   # (a cooperative LSTM kernel that gave up on a poll: the step's values are NaN)
        if step >= args.warmup:
            times.append(dt)
        if rank == 0:
            print('[TRAIN {:0>5d}] '.format(step + 1) + ' '.join('{}: {:.6f}'.format(k, v) for k, v in vals.items()) +
                  ' elapsed: {:.3f} secs'.format(dt))
    import torch.distributed as dist
    if dist.is_initialized():
        t = torch.tensor([float(np.median(times))], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        med = float(t.item())
    else:
        med = float(np.median(times))
    if rank == 0:
        res = {'steps_per_sec': 1.0 / med, 'frames_per_sec': world * B * F / med, 'n_gpus': world,
               'windows_per_gpu': B, 'window_size': F, 'median_step_ms': med * 1e3, 'hip_graph': bool(args.graph),
               'process_group': dist.get_backend() if dist.is_initialized() else None,
               'gradient_collectives_per_step': n_coll,
               'host_enqueue_ms_median': float(np.median(enq[args.warmup:]) * 1e3) if enq else None}
        print(json.dumps(res) if args.json else res)
    import torch.distributed as dist
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
