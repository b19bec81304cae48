#!/usr/bin/env python
"""
Evaluate an end-to-end model on the EM-POSE recordings (mirror of reference scripts/evaluate_real.py:24-110).

    python scripts/evaluate_real.py --model_id 1615631737 [--cross_subject]

needs the assets the reference needs (EM_EXPERIMENTS, EM_DATA_REAL, SMPL_MODELS: released weights, recordings and the
licensed SMPL-H model); they cannot be shipped with this repository.  Without them,

    python scripts/evaluate_real.py --synthetic [--n_markers 6 --iterations 2]

runs the same driver on the stand-in of BASELINE.json configs[3]: 36 synthetic recordings with the frame counts of
the real test set (README.md:107-142, 54 030 frames), a synthetic SMPL-H-shaped body model and random-init weights of
the released LGD-RNN-6 architecture, streamed in 256-frame chunks with LSTM state carry.

BASELINE.json configs[0] (plumbing, no GPU): the frame-wise ResNet baseline runs as plain PyTorch on CPU tensors, like
the reference's `C.DEVICE` fallback (reference configuration.py:23, scripts/evaluate_real.py:24-61):

    python scripts/evaluate_real.py --synthetic --m_type resnet --device cpu --max_sequences 1

Only that model has a CPU path (the LGD models are the HIP path and refuse CPU tensors); without a device the joint
position metrics, which need an SMPL-H evaluation, are skipped and the joint-angle metric is reported.

Multi-GPU: `--gpus N` (spawns one rank per GPU; `python -m torch.distributed.run --nproc-per-node N
scripts/evaluate_real.py ...` works too); whole recordings are assigned to ranks longest-first (chunks of one
recording are serially dependent), and the per-rank metric accumulators are combined with one all_gather over RCCL.
"""
import argparse
import glob
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
from tabulate import tabulate  # noqa: E402

from em_pose_amd.data.data import RealBatch, RealSample  # noqa: E402
from em_pose_amd.data.transforms import NormalizeRealMarkers, NormalizeRoot, ToTensor  # noqa: E402
from em_pose_amd.eval.helpers import (evaluate_sequences, evaluate_sequences_batched, load_model,  # noqa: E402
                                      partition_sequences)
from em_pose_amd.helpers.configuration import CONSTANTS as C  # noqa: E402


def sample_to_batch(sample):
    sample = ToTensor()(NormalizeRealMarkers()(sample))
    return NormalizeRoot()(RealBatch.from_sample_list([sample]))


def _hip_sensors(net, device):
    def sensors(poses, betas, o_r, o_t):
        n = poses.shape[0]
        pos, ori, _ = net.get_estimated_real_markers(torch.from_numpy(poses).to(device),
                                                     torch.from_numpy(betas).to(device),
                                                     torch.from_numpy(o_r[:1].copy()).to(device),
                                                     torch.from_numpy(o_t[:1].copy()).to(device), frames_per_window=n)
        return pos.cpu().numpy(), ori.cpu().numpy()
    return sensors


def synthetic_setup(args, device):
    from em_pose_amd import synthetic
    from em_pose_amd.bodymodels.smpl import SMPLLayer
    from em_pose_amd.helpers.configuration import Configuration, lgd_config
    from em_pose_amd.nn.models import create_model
    torch.manual_seed(args.model_id)
    if args.m_type == 'resnet':
        # the frame-wise baseline (reference models.py:166-262); hyper-parameters of the released ResNet are not
        # published (its config.json is part of models.zip), so a 3-block, 512-wide stand-in
        cfg = Configuration.defaults(m_type='resnet', m_hidden_size=512, m_num_layers=3, use_marker_pos=True,
                                     use_marker_ori=True, n_markers=args.n_markers, window_size=32,
                                     m_estimate_shape=False, m_fk_loss=0.0)
        net = create_model(cfg, None).to(device).eval()
        if device.type == 'cpu':
            # no device, no body model: sensor readings of the stand-in recordings are random (plumbing only)
            smpl = None
            rng = np.random.default_rng(args.model_id)

            def sensors(poses, betas, o_r, o_t):
                n = poses.shape[0]
                ori = synthetic._exp_so3(rng.normal(0, 0.5, size=(n, 12, 3))).astype(np.float32)
                return rng.normal(0, 0.3, size=(n, 12, 3)).astype(np.float32), ori
        else:
            smpl = SMPLLayer(synthetic.make_model()).to(device)
            lgd = create_model(lgd_config(12, False, 1, hidden=32), SMPLLayer(synthetic.make_model())).to(device).eval()
            sensors = _hip_sensors(lgd, device)
    else:
        model = synthetic.make_model()
        cfg = lgd_config(args.n_markers, not args.no_rnn, args.iterations, lr=0.0005)
        net = create_model(cfg, SMPLLayer(model)).to(device).eval()
        smpl = SMPLLayer(model).to(device)
        sensors = _hip_sensors(net, device)

    lengths = synthetic.README_SEQUENCE_LENGTHS[:args.max_sequences] if args.max_sequences else \
        synthetic.README_SEQUENCE_LENGTHS

    def load(i):
        d = synthetic.make_sequence(lengths[i], 100 + i, sensors)
        s = RealSample(str(d['id']), d['sensor_pos'], d['sensor_oris'], d['sensor_masks'].astype(np.float32),
                       d['smpl_poses'], d['smpl_shape'], d['smpl_trans'],
                       {'means': d['offset_means'], 'covs': d['offset_covs'], 'r': d['offset_r']})
        return sample_to_batch(s)
    return net, smpl, lengths, load, 'synthetic-%d' % args.model_id


def real_setup(args, device):
    from em_pose_amd.bodymodels.smpl import create_default_smpl_model
    for var in ('EM_EXPERIMENTS', 'EM_DATA_REAL', 'SMPL_MODELS'):
        if not os.environ.get(var):
            raise SystemExit('{} is not set. The real evaluation needs the released weights, the recordings and the '
                             'licensed SMPL-H model; use --synthetic to run the driver without them.'.format(var))
    net, config, _ = load_model(args.model_id, device)
    smpl = create_default_smpl_model(device)
    base = os.path.join(C.DATA_DIR_TEST, 'hold_out') if args.cross_subject else C.DATA_DIR_TEST
    files = sorted(glob.glob(os.path.join(base, '*_clean.npz')))
    if not files:
        raise SystemExit('no *_clean.npz recordings under ' + base)
    lengths = [int(np.load(f)['smpl_poses'].shape[0]) for f in files]
    return net, smpl, lengths, (lambda i: sample_to_batch(RealSample.from_npz_clean(files[i]))), str(args.model_id)


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--model_id', type=int, default=1615631737, help='Which end-to-end model to evaluate.')
    p.add_argument('--visualize', type=int, default=-1, help='(accepted for compatibility; not implemented)')
    p.add_argument('--cross_subject', action='store_true', help='Evaluate on hold-out subject 0715.')
    p.add_argument('--synthetic', action='store_true', help='Synthetic recordings / body model / weights.')
    p.add_argument('--n_markers', type=int, default=6)
    p.add_argument('--iterations', type=int, default=2)
    p.add_argument('--no_rnn', action='store_true')
    p.add_argument('--m_type', default='ief', choices=['ief', 'resnet'], help='--synthetic only: which model to build.')
    p.add_argument('--device', default='cuda', choices=['cuda', 'cpu'],
                   help="'cpu' runs the ResNet baseline as plain PyTorch (BASELINE configs[0]); the LGD models need 'cuda'.")
    p.add_argument('--max_sequences', type=int, default=0)
    p.add_argument('--json', action='store_true', help='Also print one machine-readable JSON line.')
    p.add_argument('--sequential', action='store_true',
                   help='One recording at a time like the reference; default: chunk c of all recordings as one batch.')
    p.add_argument('--repeat', type=int, default=1,
                   help='Evaluate the set this many times and report every pass; the first pass still pays one-time '
                        'costs (kernels and allocations of every batch shape that occurs), later ones do not.')
    p.add_argument('--gpus', type=int, default=1,
                   help='Shard whole recordings over this many GPUs of the node (spawns one rank per GPU).')
    p.add_argument('--force_dist', action='store_true',
                   help='Initialise the process group (RCCL) and run the metric gather even with one rank (self-test).')
    p.add_argument('--no_warmup', action='store_true',
                   help='Do not run the first chunk of the first recording once before the timed evaluation '
                        '(code-object loading, workspace allocation).')
    args = p.parse_args()

    # `--gpus N` as a plain command spawns its own N ranks (helpers/distributed.py); under torch.distributed.run the ranks
    # exist already.  --force_dist sends one rank through the same spawn + RCCL path (single-GPU self-test).
    from em_pose_amd.helpers import distributed as D
    on_cpu = args.device == 'cpu'
    if on_cpu and (args.gpus > 1 or not (args.synthetic and args.m_type == 'resnet')):
        raise SystemExit("--device cpu is the single-process plumbing configuration of the ResNet baseline "
                         "(--synthetic --m_type resnet); the LGD models run the HIP path and need an MI355X.")
    if not on_cpu:
        D.maybe_self_launch(os.path.abspath(__file__), args.gpus, force=args.force_dist, what='evaluate_real.py')
    launched = D.launched_by_a_launcher()
    world = int(os.environ.get('WORLD_SIZE', '1')) if launched else 1
    rank = int(os.environ.get('RANK', '0')) if launched else 0
    local_rank = int(os.environ.get('LOCAL_RANK', '0')) if launched else 0
    if args.gpus not in (1, world):
        raise SystemExit('--gpus {} but WORLD_SIZE={} (RANK / WORLD_SIZE inherited from the environment, e.g. a SLURM step or '
                         'a parent launcher\'s shell, count as "already launched": unset them or pass the matching --gpus)'
                         .format(args.gpus, world))
    dist = None
    if on_cpu:
        if world > 1:
            raise SystemExit('--device cpu is a single-process configuration')
        device = torch.device('cpu')
    else:
        if launched and D.dist_backend(torch.device('cuda')) == 'gloo':   # device-less rendezvous: the launcher's CPU test
            dist, rank, world = D.init_process_group(None, log=lambda m: print(m, file=sys.stderr, flush=True))
            dist.barrier()
        if not torch.cuda.is_available():
            raise SystemExit('evaluate_real.py runs the HIP path and needs an MI355X; there is no CPU fallback '
                             '(the ResNet plumbing configuration: --synthetic --m_type resnet --device cpu).')
        device = torch.device('cuda', D.local_device_index(local_rank, 'evaluate_real.py'))
        torch.cuda.set_device(device)
    sync = (lambda: None) if on_cpu else torch.cuda.synchronize
    if dist is None and launched and (world > 1 or args.force_dist) and not on_cpu:
        dist, rank, world = D.init_process_group(device)

    net, smpl, lengths, load, name = (synthetic_setup if args.synthetic else real_setup)(args, device)
    mine = partition_sequences(lengths, world)[rank]
    batches = [load(i) for i in mine]  # data preparation is outside the timed region
    if not on_cpu:   # what DataLoader(pin_memory=True) does with a batch: uploads neither stage nor block the host
        batches = [b.pin_memory() for b in batches]
    net.keep_history = False
    from em_pose_amd.nn.models import IterativeErrorFeedback
    sequential = args.sequential or not isinstance(net, IterativeErrorFeedback)
    if batches and not args.no_warmup:   # the first 256-frame chunk, through the driver that is about to be timed
        from em_pose_amd.eval.helpers import window_generator
        if sequential:
            evaluate_sequences(net, [next(iter(window_generator(batches[0], 256)))], smpl, device, window_size=256)
        else:
            evaluate_sequences_batched(net, [next(iter(window_generator(b, 256))) for b in batches], smpl, device,
                                       window_size=256)
    sync()
    if dist is not None:
        dist.barrier()
    log = print if world == 1 else None
    passes, host_times = [], []
    for rep in range(max(args.repeat, 1)):
        t0 = time.perf_counter()
        if sequential:
            ht = {}
            me_all, per_seq, frames = evaluate_sequences(net, batches, smpl, device, window_size=256,
                                                         log=log if rep == 0 else None, host_times=ht)
            host_times.append({k: round(v, 4) for k, v in ht.items()})
        else:
            ht = {}
            me_all, per_seq, frames = evaluate_sequences_batched(net, batches, smpl, device, window_size=256,
                                                                 host_times=ht)
            host_times.append({k: round(v, 4) for k, v in ht.items()})
        sync()
        passes.append(time.perf_counter() - t0)
    # what is reported: the median of the warm passes (pass 0 still pays for every batch shape that occurs for the first
    # time); all passes are listed in `seconds_per_pass`
    elapsed = float(np.median(passes[1:])) if len(passes) > 1 else passes[0]
    rows = [(i, sid, m) for i, (sid, m) in zip(mine, per_seq)]
    if dist is not None:
        me_all.gather(device=D.collective_device(device), force=args.force_dist)
        gathered = [None] * world
        dist.all_gather_object(gathered, (rows, frames, elapsed, passes))
        rows = sorted(r for g in gathered for r in g[0])
        frames = sum(g[1] for g in gathered)
        elapsed = max(g[2] for g in gathered)
        passes = [max(g[3][k] for g in gathered) for k in range(len(passes))]
    if rank == 0:
        metrics = me_all.get_metrics()
        table = [[i, sid] + list(m.values()) for i, sid, m in rows]
        table.append([len(table), 'Overall average'] + list(metrics.values()))
        print(tabulate(table, headers=['Nr', 'E2E {}'.format(name)] + list(metrics.keys())))
        print('{} frames in {:.3f} s on {}: {:.0f} frames/s (model forward + metrics)'
              .format(frames, elapsed, 'the host CPU' if on_cpu else '{} GPU(s)'.format(world), frames / elapsed))
        if args.json:
            print(json.dumps({'frames': frames, 'seconds': elapsed, 'n_gpus': world, 'frames_per_sec': frames / elapsed,
                              'seconds_per_pass': passes, 'host_seconds_per_section_per_pass': host_times,
                              'metrics': metrics, 'headers': list(metrics.keys()),
                              'rows': [[r[0], str(r[1])] + [float(v) for v in r[2:]] for r in table]}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
