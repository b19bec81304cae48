#!/usr/bin/env python
"""
Headline benchmark: frames/sec of the LGD-RNN-12 inference forward (N=4 iterations, window size 32, 12 sensors,
batch 1024 windows per GPU -- BASELINE.json configs[2], model 1615200973's architecture) on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N>1: spawns its own N ranks, one per GPU; also runs as the
                                                          ranks of `python -m torch.distributed.run --nproc-per-node N`)

A step is one call of IterativeErrorFeedback.forward_tensors -> empose_lgd_forward on a device-resident batch of
synthetic 12-sensor windows (synthetic SMPL-H-shaped body model, V=6890; random-init weights of the released
architecture, BatchNorm statistics randomised; the licensed SMPL-H asset and the released weights cannot be shipped).
Windows are independent, so ranks shard them with no data-path collective (weak scaling); the only RCCL traffic is an
all-gather of per-rank result checksums / counts at the end.

Rank 0 prints ONE JSON line. Besides the contract fields it carries
  roofline      the dominant kernel (mlp_fused_kernel: both update MLPs, all layers, one launch per LGD iteration; the
                hidden-layer GEMM launch when the batch is too small for it) against the fp32 matrix-core peak, from
                HIP events recorded around every launch on the launch stream (empose_profile_*), in a profiling pass
                of the same workload right after the timed region;
  cpu_baseline  the oracle (oracle/torch_ref.py: dense full-mesh SMPL-H + autograd, the reference's algorithm) timed on
                the host cores on a bounded sample of the same workload (rank 0, N=1 only); the sample's outputs are
                also compared with the HIP outputs for the same windows: `mpjpe_hip_vs_oracle_mm` (the "MPJPE vs ref"
                half of the metric) and `max_abs_diff_pose_shape_joints` (north-star tolerance: 1e-4).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 4 SIMDs x 64 FLOP/clk x 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 dense (v_mfma_f32_32x32x16_bf16: 32 cycles per SIMD)
X3_PRODUCTS = 6                 # bf16 piece products per fp32 product in mlp_fused_x3.hip (three pieces per operand)
PEAK_HBM_GBS = 8000.0
MODEL_SEED = 1615200973  # the released LGD-RNN-12 model id (BASELINE.json configs[2])


def build_net(n_markers, rnn, N):
    from em_pose_amd import synthetic
    from em_pose_amd.bodymodels.smpl import SMPLLayer
    from em_pose_amd.helpers.configuration import lgd_config
    from em_pose_amd.nn.models import create_model
    model = synthetic.make_model()
    torch.manual_seed(MODEL_SEED)
    net = create_model(lgd_config(n_markers, rnn, N), SMPLLayer(model))
    g = torch.Generator().manual_seed(MODEL_SEED + 1)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
    return net.eval(), model


def make_inputs(net, dev, B, F, seed):
    """Synthetic windows; the sensor readings come from the HIP body model itself (+ 5 mm / 2 deg noise)."""
    from em_pose_amd import synthetic

    def sensors(poses, betas, o_r, o_t):
        T = poses.shape[0]
        pos, ori, _ = net.get_estimated_real_markers(torch.from_numpy(poses).to(dev), torch.from_numpy(betas).to(dev),
                                                     torch.from_numpy(o_r[::F].copy()).to(dev),
                                                     torch.from_numpy(o_t[::F].copy()).to(dev), frames_per_window=F)
        return pos.cpu().numpy(), ori.cpu().numpy()
    w = synthetic.make_windows(B, F, seed, sensors)
    return w, [torch.from_numpy(w[k]).to(dev) for k in ('marker_pos', 'marker_oris', 'offset_t', 'offset_r')]


def flops_per_frame(net):
    """Dense-contraction flops of one frame: LSTM + heads + N x two 6-layer MLPs + blend matrices."""
    d_in, dx, N = net.input_size, net.input_iter_size, net.N
    h = net.config.m_hidden_size
    mlp = lambda i, o: 2 * (i * h + 4 * h * h + h * o)
    fl = N * (mlp(dx, 66) + mlp(dx, 10))
    if net.rnn_init:
        H, L = net.rnn.hidden_size, net.rnn.num_layers
        fl += 2 * 4 * H * (d_in + H) + (L - 1) * 2 * 4 * H * (H + H) + 2 * H * 76
    else:
        fl += mlp(d_in, 66) + mlp(d_in, 10)
    return fl


def arithmetic_label(x3_options, mlp_x3_on, rnn):
    """Which products of the headline ran as three-piece bf16 (fp32-equivalent) and which on the fp32 MFMA instruction,
    from the library's options at the time of the timed region."""
    on, off = [], []
    (on if mlp_x3_on else off).append('update MLPs (mlp_fused%s.hip)' % ('_x3' if mlp_x3_on else ''))
    if rnn:
        (on if x3_options[b'lstm_x3'] else off).append('LSTM steps (lstm%s.hip)' % ('_x3' if x3_options[b'lstm_x3'] else ''))
    (on if x3_options[b'rows_x3'] else off).append('init heads and SMPL blend GEMMs (rows kernels of mlp_fused.hip)')
    txt = 'fp32 operands and fp32 accumulation'
    if on:
        txt += ('; every fp32 product of the %s is formed from three bf16 pieces per operand (8+8+8 mantissa bits) as six bf16 '
                'matrix-core products: fp32-equivalent, error against float64 measured equal to the fp32 MFMA '
                'instruction\'s (tests/test_hip_round5.py)' % ', the '.join(on))
    if off:
        txt += '; %s on the fp32 MFMA instruction' % ', '.join(off)
    return txt


def baseline_config_label(net, n_markers, F, B):
    """Which entry of BASELINE.json `configs` the command-line shape is (the headline is configs[2])."""
    if n_markers == 12 and net.N == 4 and F == 32:
        if net.rnn_init and B == 1024:
            return 'BASELINE configs[2]'
        if not net.rnn_init and B == 256:
            return 'BASELINE configs[1]'
    return 'custom shape (not a BASELINE config)'


def cpu_baseline(net, model, w, hip_out=None, seconds_target=20.0):
    """Oracle (reference algorithm: dense full mesh + autograd) on the host cores, bounded sample.  The sample's outputs
    are also compared with what the HIP path produced for the same windows (`hip_out`): the "MPJPE vs ref" half of the
    metric, in mm, and the largest absolute difference over pose / shape / joints."""
    from em_pose_amd.helpers.configuration import CONSTANTS as C
    from oracle import torch_ref as R
    bm = R.BodyModelTensors(model)
    tables = R.sensor_tables(model['f'], C.VERTEX_IDS)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items() if not k.startswith('smpl.')}
    F = w['marker_pos'].shape[1]

    def run(nb):
        inp = {k: torch.from_numpy(w[k][:nb]) for k in ('marker_pos', 'marker_oris', 'offset_t', 'offset_r')}
        inp['marker_masks'] = None
        inp['seq_lengths'] = torch.full((nb,), F, dtype=torch.int64)
        t0 = time.perf_counter()
        last['out'], _ = R.ief_forward(sd, bm, tables, C.VERTEX_IDS, inp, n_markers=net.n_markers, N=net.N,
                                       rnn_init=net.rnn_init)
        return time.perf_counter() - t0
    last = {}

    run(2)  # warm-up (thread pools, allocator)
    # The oracle is thousands of small torch ops: more threads are not faster (128 threads measured slower than one).
    # A cheap sweep over thread counts on 8 windows picks the best count; the reported figures are then measured on the
    # 32-window sample BASELINE.md section 3 / SURVEY 8d name: at ONE core, at ALL cores, and at the best count (= value).
    all_threads = int(torch.get_num_threads())
    nb = min(32, w['marker_pos'].shape[0])
    sweep = {}
    try:
        for n in sorted({1, 8, 32, all_threads}):
            if n > all_threads:
                continue
            torch.set_num_threads(n)
            sweep[n] = min(8, nb) * F / run(min(8, nb))
        best_n = max(sweep, key=sweep.get)
        timed = {}
        order = [n for n in dict.fromkeys((1, all_threads)) if n != best_n] + [best_n]
        for n in order:      # best_n last: its outputs are the ones compared below
            torch.set_num_threads(n)
            reps = timed.setdefault(n, [])
            reps.append(run(nb))
            while n == best_n and sum(reps) < seconds_target / 2 and len(reps) < 3:
                reps.append(run(nb))
    finally:
        torch.set_num_threads(all_threads)
    rate = lambda n: nb * F / float(np.median(timed[n]))
    parity = {}
    if hip_out is not None:
        want = last['out']
        j_ref = want['joints_hat'].reshape(nb, F, 22, 3).double()
        j_hip = hip_out['joints'][:nb].detach().cpu().reshape(nb, F, 22, 3).double()
        pose_hip = hip_out['pose'][:nb].detach().cpu()
        diffs = [(j_hip - j_ref).abs().max(), (pose_hip[..., 3:] - want['pose_hat']).abs().max(),
                 (pose_hip[..., :3] - want['root_ori_hat']).abs().max(),
                 (hip_out['shape'][:nb].detach().cpu() - want['shape_hat']).abs().max()]
        parity = {'mpjpe_hip_vs_oracle_mm': float((j_hip - j_ref).norm(dim=-1).mean() * 1000.0),
                  'max_abs_diff_pose_shape_joints': float(max(diffs))}
    return {'value': rate(best_n), 'unit': 'frames/sec', 'cores': best_n, 'kind': 'port',
            'frames_per_sec_1_core': rate(1), 'frames_per_sec_all_cores': rate(all_threads), 'all_cores': all_threads,
            **parity,
            'sample': '%d windows x %d frames through oracle/torch_ref.ief_forward (dense V=6890 SMPL-H + autograd, fp32, '
                      'torch CPU): value = median of %d runs at the best of the swept thread counts (%d); one run each at '
                      '1 thread and at all %d threads; host has %d logical cores'
                      % (nb, F, len(timed[best_n]), best_n, all_threads, os.cpu_count()),
            'threads_sweep_frames_per_sec_8_windows': {str(k): v for k, v in sweep.items()}}


def pmc_traffic_live(argv_tail, kernel_name, timeout_s=240):
    """HBM bytes per launch of the dominant kernel measured NOW: two `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE:
    separate runs, kernel trace only, as MI355X_MICROARCH.md's HBM section prescribes) of this same command with two
    timed steps, as child processes.  The counter unit is KiB; on gfx950 FETCH_SIZE reports half of the bytes of a wide
    coalesced streaming read, so bytes = (2 FETCH_SIZE + WRITE_SIZE) * 1024.  Returns (bytes, description) or
    (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(exe):
        return None, 'rocprofv3 not found'
    norm = lambda k: k.replace('empose::', '').replace('void ', '').replace(' ', '')
    means = {}
    env = dict(os.environ, TMPDIR='/tmp')
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        out = tempfile.mkdtemp(prefix='empose_pmc_', dir='/tmp')
        try:
            cmd = [exe, '--pmc', counter, '--kernel-trace', '--output-format', 'csv', '-d', out, '-o', 'pmc', '--',
                   sys.executable, os.path.abspath(__file__)] + argv_tail + \
                  ['--steps', '2', '--warmup', '1', '--no_cpu_baseline', '--no_profile', '--no_secondary']
            r = subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                               timeout=timeout_s)
            vals = []
            for f in glob.glob(os.path.join(out, '**', '*counter_collection.csv'), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if row.get('Counter_Name') == counter and norm(row['Kernel_Name']).startswith(kernel_name):
                            vals.append(float(row['Counter_Value']))
            if not vals:
                return None, 'rocprofv3 --pmc %s pass gave no rows for %s (exit code %d)' % (counter, kernel_name,
                                                                                          r.returncode)
            means[counter] = (sum(vals) / len(vals), len(vals))
        except (subprocess.TimeoutExpired, OSError) as e:
            return None, 'rocprofv3 --pmc %s pass failed: %s' % (counter, type(e).__name__)
        finally:
            shutil.rmtree(out, ignore_errors=True)
    nbytes = (2.0 * means['FETCH_SIZE'][0] + means['WRITE_SIZE'][0]) * 1024.0
    pmc_traffic_live.raw = {'FETCH_SIZE_KiB_per_launch': means['FETCH_SIZE'][0],
                            'WRITE_SIZE_KiB_per_launch': means['WRITE_SIZE'][0],
                            'formula': 'bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md, HBM [CDNA4]: on '
                                       'gfx950 FETCH_SIZE reports half of a wide coalesced read; WRITE_SIZE uncalibrated)'}
    return nbytes, ('measured by this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE child passes of this command '
                    '(2 timed steps each; %d / %d dispatches of the kernel), bytes = (2 FETCH_SIZE + WRITE_SIZE) KiB'
                    % (means['FETCH_SIZE'][1], means['WRITE_SIZE'][1]))


def pmc_traffic(T, hidden, kernel_name):
    """HBM bytes per launch of the dominant kernel, NOT measured in this run: hardware counters need rocprofv3 around the
    process (`rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of this same command, scripts/dev/run_pmc.sh,
    corrected as MI355X_MICROARCH.md prescribes), so the number is read from the newest committed
    profiles/*pmc_hbm_traffic.json and reported together with that file's name. Only valid for the workload the counters
    were collected on; otherwise (None, None)."""
    import glob
    if (T, hidden) != (32768, 512):
        return None, None
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*pmc_hbm_traffic.json')))
    if not files:
        return None, None
    with open(files[-1]) as f:
        ks = json.load(f)['kernels']
    norm = lambda k: k.replace('empose::', '').replace('void ', '').replace(' ', '')
    hit = [v for k, v in ks.items() if norm(k).startswith(kernel_name.replace(' ', ''))]
    if not hit:
        return None, None
    return hit[0]['hbm_bytes_per_launch_corrected'], 'profiles/' + os.path.basename(files[-1])


def vertices_numbers(smpl, dev, T, steps, warmup):
    """Stand-alone full-mesh SMPL-H evaluation `smpl_vertices_fwd` (SMPLLayer.forward -> empose_mesh_vertices_fwd) on T
    frames: wall-clock frames/s, device ms per call, and the two roofline fractions on SURVEY.md 8(d)'s per-frame figures."""
    g = torch.Generator().manual_seed(3)
    pose = (torch.randn(T, 63, generator=g) * 0.3).to(dev)
    root = (torch.randn(T, 3, generator=g) * 0.3).to(dev)
    betas = torch.randn(T, 10, generator=g).to(dev)
    for _ in range(warmup):
        v, j = smpl(poses_body=pose, betas=betas, poses_root=root)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        v, j = smpl(poses_body=pose, betas=betas, poses_root=root)
    e1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    dev_ms = e0.elapsed_time(e1) / steps
    V = smpl.n_vertices
    bytes_frame = 66 * 4 + 10 * 4 + V * 3 * 4 + 66 * 4
    flops_frame = 2.0 * 200 * (V * 3 + 66) + V * 3 * (4 * 8 + 8)
    return {'fps': T * steps / wall, 'wall_ms': 1000.0 * wall / steps, 'dev_ms': dev_ms, 'V': V,
            'bytes_frame': bytes_frame, 'flops_frame': flops_frame,
            'hbm': bytes_frame * T / (dev_ms * 1e-3) / 1e9, 'tfl': flops_frame * T / (dev_ms * 1e-3) / 1e12,
            'tfl_contraction': 2.0 * 200 * (V * 3 + 66) * T / (dev_ms * 1e-3) / 1e12}


def vertices_arithmetic(arith):
    """(label, kernel name, True when the contraction runs as bf16 piece products) of the full-mesh path now selected."""
    from em_pose_amd import _lib
    if arith == 'bf16x3':
        return ('two-piece split bf16 on the pose columns (3 products), three pieces on the shape columns: NOT fp32-equivalent, '
                'explicitly selected', 'mesh_rows_bf16_kernel', True)
    opt = _lib.lib().empose_get_option(b'mesh_x3')
    if opt != 0:
        return ('fp32 operands and fp32 accumulation; every fp32 product of the blend-shape contraction is formed from three '
                'bf16 pieces per operand as six bf16 matrix-core products (mesh_x3.hip): fp32-equivalent, error against float64 '
                'measured equal to the fp32 MFMA kernel\'s (tests/test_hip_round6.py); chain and skinning fp32 vector '
                'arithmetic', 'mesh_rows_x3_kernel', True)
    return 'fp32 MFMA instruction (mesh_rows_kernel), fp32 vector skinning', 'mesh_rows_kernel', False


def run_vertices(args, dev):
    """Secondary workload: stand-alone full-mesh SMPL-H evaluation `smpl_vertices_fwd`
    (SMPLLayer.forward -> empose_mesh_vertices_fwd), the only piece of the path whose algorithmic traffic is large:
    83 842 B/frame (82 680 B of vertices written).  Its roofline is the HBM one (north_star: >= 40 % asked for THIS kernel)."""
    from em_pose_amd import synthetic
    from em_pose_amd.bodymodels.smpl import SMPLLayer
    smpl = SMPLLayer(synthetic.make_model(), arithmetic=args.arith).to(dev)
    T = args.batch * args.frames
    n = vertices_numbers(smpl, dev, T, args.steps, args.warmup)
    label, kname, pieces = vertices_arithmetic(args.arith)
    traffic, traffic_src = None, 'skipped (--no_traffic)'
    if not args.no_traffic:
        tail = ['--workload', 'vertices', '--arith', args.arith, '--batch', str(args.batch), '--frames', str(args.frames),
                '--no_traffic'] + [a for kv in args.option for a in ('--option', kv)]
        traffic, traffic_src = pmc_traffic_live(tail, kname)
    print(json.dumps({
        'metric': 'frames/sec SMPL-H full-mesh vertices (smpl_vertices_fwd%s)' % (', two-piece split-bf16 variant'
                                                                                 if args.arith == 'bf16x3' else ''),
        'value': n['fps'], 'unit': 'frames/sec',
        'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': n['wall_ms'],
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'SMPLLayer.forward: %d frames per step, V=%d vertices + 52 joints, synthetic SMPL-H-shaped '
                               'model' % (T, n['V']), 'frames_per_step': T, 'arithmetic': label},
        'roofline': {'bound': 'hbm', 'achieved': n['hbm'], 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                     'frac': n['hbm'] / PEAK_HBM_GBS, 'traffic': traffic, 'traffic_source': traffic_src,
                     'traffic_raw_counters': getattr(pmc_traffic_live, 'raw', None) if traffic is not None else None,
                     'kernel': 'update_feat + rest-joint gemm + mesh_chain + %s (device time of the call; the last one is '
                               '> 95 %% of it)' % kname,
                     'device_ms_per_step': n['dev_ms'], 'hbm_GBs_on_algorithmic_bytes': n['hbm'],
                     'hbm_frac': n['hbm'] / PEAK_HBM_GBS, 'algorithmic_TFLOPs': n['tfl'],
                     'executed_bf16_matrix_TFLOPs': n['tfl_contraction'] * X3_PRODUCTS if pieces and args.arith != 'bf16x3' else None,
                     'mfma_frac_of_its_roof': (n['tfl_contraction'] * X3_PRODUCTS / PEAK_BF16_MFMA_TFLOPS)
                     if pieces and args.arith != 'bf16x3' else n['tfl_contraction'] / PEAK_FP32_MFMA_TFLOPS,
                     'algorithmic_bytes_per_frame': n['bytes_frame'], 'flops_per_frame': n['flops_frame'],
                     'what_binds': 'not HBM: the six bf16 products of the blend shapes (0.5 ms of 1.06 at 16384 frames), the '
                                   'LDS-bound gather of four bone transforms per vertex and frame (0.35 ms) and the '
                                   'staggered stores; profiles/r06_mesh_x3_lab.txt'}}))


def _json_tail(cmd, timeout_s):
    """Last stdout line of a child command as JSON, or {'error': ...}."""
    import subprocess
    try:
        r = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=timeout_s)
        lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip().startswith('{')]
        return json.loads(lines[-1]) if lines else {'error': 'no JSON line (exit code %d)' % r.returncode}
    except Exception as e:    # noqa: BLE001 (a secondary number must never take the headline down)
        return {'error': '%s: %s' % (type(e).__name__, e)}


def secondary_workloads(args, dev, model):
    """The other BASELINE configs and the full-mesh workload, measured after the headline on the same GPU, so that they are
    visible to whoever reads the ONE result line.  Never `value`; each entry says what it ran.
      configs[1]: LGD-12 without the RNN, N = 4, 256 windows, in this process (frames/s + dominant-kernel fraction)
      vertices:   smpl_vertices_fwd at 16384 frames, in this process (frames/s + HBM fraction on 83 842 B/frame)
      configs[3]: scripts/evaluate_real.py --synthetic (36 sequences with the README's lengths, 54 030 frames, LGD-RNN-6
                  N = 2, 256-frame chunks with carried state), batched and sequential drivers, child processes
      configs[4]: scripts/train.py synthetic step (LGD-RNN-12, N = 4, forward + backward + Adam) at 12 and 256 windows of
                  32 frames, child processes; training roofline = 3 x the forward's dense flops / step time / fp32 MFMA peak"""
    from em_pose_amd import _lib
    from em_pose_amd.bodymodels.smpl import SMPLLayer
    lib = _lib.lib()
    sec = {}
    t_all = time.perf_counter()
    # ---- configs[1]
    try:
        net, _ = build_net(12, False, 4)
        net = net.to(dev)
        B, F = 256, 32
        _, inputs = make_inputs(net, dev, B, F, seed=2000)
        for _ in range(3):
            net.forward_tensors(*inputs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            out = net.forward_tensors(*inputs)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert torch.isfinite(out['pose']).all()
        entry = {'workload': 'BASELINE configs[1]: LGD (no RNN) 12-sensor, N=4, 2x512 MLP, batch 256 windows x 32 frames',
                 'frames_per_sec': B * F * 20 / dt, 'ms_per_step': 1000.0 * dt / 20}
        _lib.check(lib.empose_profile_enable_only(b'mlp_fused'))
        for _ in range(10):
            net.forward_tensors(*inputs)
        solo = _lib.profile_read()
        lib.empose_profile_enable(0)
        if 'mlp_fused' in solo:
            avg_ms = solo['mlp_fused'][0] / solo['mlp_fused'][1]
            flops = sum(2.0 * (B * F) * lin.in_features * lin.out_features
                        for mlp in (net.pose_net_iter, net.shape_net_iter) for lin, _, _ in mlp.dense_specs())
            x3 = lib.empose_get_option(b'mlp_x3') != 0
            peak = PEAK_BF16_MFMA_TFLOPS / X3_PRODUCTS if x3 else PEAK_FP32_MFMA_TFLOPS
            entry['roofline'] = {'bound': 'mfma', 'kernel': 'mlp_fused_x3_kernel' if x3 else 'mlp_fused_kernel',
                                 'avg_launch_ms': avg_ms, 'achieved': flops / (avg_ms * 1e-3) / 1e12, 'peak': peak,
                                 'unit': 'TFLOP/s', 'frac': flops / (avg_ms * 1e-3) / 1e12 / peak}
        sec['configs1_lgd12_b256'] = entry
        del net, inputs
    except Exception as e:    # noqa: BLE001
        sec['configs1_lgd12_b256'] = {'error': '%s: %s' % (type(e).__name__, e)}
    # ---- vertices
    try:
        smpl = SMPLLayer(model).to(dev)
        n = vertices_numbers(smpl, dev, 16384, 40, 8)    # (the clocks settle over the first few calls of a 1 ms workload)
        label, kname, _ = vertices_arithmetic('f32')
        sec['vertices_t16384'] = {'workload': 'smpl_vertices_fwd (SMPLLayer.forward), 16384 frames, V=%d' % n['V'],
                                  'frames_per_sec': n['fps'], 'device_ms_per_step': n['dev_ms'], 'kernel': kname,
                                  'roofline': {'bound': 'hbm', 'achieved': n['hbm'], 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                                               'frac': n['hbm'] / PEAK_HBM_GBS,
                                               'algorithmic_bytes_per_frame': n['bytes_frame']},
                                  'arithmetic': label}
        del smpl
    except Exception as e:    # noqa: BLE001
        sec['vertices_t16384'] = {'error': '%s: %s' % (type(e).__name__, e)}
    torch.cuda.empty_cache()
    # ---- configs[3] stand-in and configs[4], as child processes (their own entry points)
    py = sys.executable
    for key, extra in (('batched', []), ('sequential', ['--sequential'])):
        r = _json_tail([py, os.path.join(ROOT, 'scripts', 'evaluate_real.py'), '--synthetic', '--repeat', '8', '--json'] + extra, 180)
        sec['configs3_evaluate_real_synthetic_' + key] = r if 'error' in r else {
            'workload': 'BASELINE configs[3] stand-in: scripts/evaluate_real.py --synthetic%s (36 sequences, 54 030 frames, '
                        'LGD-RNN-6 N=2, 256-frame chunks, carried state), 1 GPU' % (' --sequential' if extra else ''),
            'frames_per_sec': r.get('frames_per_sec'), 'seconds_per_pass': r.get('seconds'), 'frames': r.get('frames'),
            'all_passes_seconds': r.get('seconds_per_pass'),
            'host_seconds_per_section_last_pass': (r.get('host_seconds_per_section_per_pass') or [None])[-1]}
    fwd_flops = 26.47e6       # dense MFLOP per frame of LGD-RNN-12 N=4; fwd + dX + dW = 3 x
    for key, extra, windows in (('12_windows', ['--steps', '20'], 12), ('256_windows', ['--steps', '8', '--bs_train', '256'], 256)):
        r = _json_tail([py, os.path.join(ROOT, 'scripts', 'train.py'), '--json'] + extra, 180)
        if 'error' in r:
            sec['configs4_train_step_' + key] = r
            continue
        ms = r.get('median_step_ms')
        tfl = 3.0 * fwd_flops * windows * 32 / (ms * 1e-3) / 1e12 if ms else None
        sec['configs4_train_step_' + key] = {
            'workload': 'BASELINE configs[4] per-GPU step: scripts/train.py synthetic, LGD-RNN-12 N=4, %d windows x 32 frames, '
                        'forward + backward + Adam, 1 GPU%s' % (windows, ', replayed as a HIP graph' if r.get('hip_graph') else ''),
            'median_step_ms': ms, 'frames_per_sec': r.get('frames_per_sec'), 'steps_per_sec': r.get('steps_per_sec'),
            'roofline': {'bound': 'mfma', 'achieved': tfl, 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': tfl / PEAK_FP32_MFMA_TFLOPS if tfl else None,
                         'flops': '3 x 26.47 MFLOP/frame (forward dense flops + the same again for dX and for dW); the GEMMs of '
                                  'the training kernels run on the fp32 MFMA instruction'}}
    sec['seconds_spent'] = time.perf_counter() - t_all
    return sec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=1024, help='windows per GPU')
    ap.add_argument('--frames', type=int, default=32)
    ap.add_argument('--n_markers', type=int, default=12)
    ap.add_argument('--iterations', type=int, default=4)
    ap.add_argument('--no_rnn', action='store_true')
    ap.add_argument('--workload', default='lgd', choices=['lgd', 'vertices'],
                    help="'lgd' = the headline LGD forward; 'vertices' = stand-alone full-mesh SMPL-H evaluation")
    ap.add_argument('--arith', default='f32', choices=['f32', 'bf16x3'],
                    help="vertices workload only: 'bf16x3' = the explicitly labelled split-bf16 variant of the blend-shape "
                         "contraction (not the reference's arithmetic; the headline never uses it)")
    ap.add_argument('--no_cpu_baseline', action='store_true')
    ap.add_argument('--no_profile', action='store_true')
    ap.add_argument('--no_fp32_line', action='store_true',
                    help='skip the second timed region on the fp32 MFMA instruction (option mlp_x3=0)')
    ap.add_argument('--no_secondary', action='store_true',
                    help='skip the `secondary` object (the other BASELINE configs + the full-mesh workload, ~1 min)')
    ap.add_argument('--no_traffic', action='store_true',
                    help='skip the two rocprofv3 --pmc child passes that measure roofline.traffic (~25 s each)')
    ap.add_argument('--force_dist', action='store_true', help='init the process group even for one rank (self-test)')
    ap.add_argument('--option', action='append', default=[], metavar='NAME=INT',
                    help='kernel-variant switch of the library (empose_set_option), for A/B runs; repeatable')
    args = ap.parse_args()

    # `python bench.py --gpus N` as a plain command spawns its own N ranks (one per GPU, RCCL over 127.0.0.1); under
    # `python -m torch.distributed.run` the ranks already exist and this returns at once.  --force_dist sends even one
    # rank through the same spawn + process-group path (single-GPU self-test of the multi-GPU entry).
    from em_pose_amd.helpers import distributed as D
    if args.workload == 'vertices' and args.gpus > 1:
        raise SystemExit('the vertices workload is a single-GPU micro-benchmark')
    D.maybe_self_launch(os.path.abspath(__file__), args.gpus, force=args.force_dist, what='bench.py')
    launched = D.launched_by_a_launcher()
    # in a rank process nothing but the result line may reach stdout (RCCL prints a version banner there)
    results_out = D.results_stream() if launched else sys.stdout
    world = int(os.environ.get('WORLD_SIZE', '1')) if launched else 1
    rank = int(os.environ.get('RANK', '0')) if launched else 0
    local_rank = int(os.environ.get('LOCAL_RANK', '0')) if launched else 0
    if args.gpus != world:
        raise SystemExit('--gpus {} but WORLD_SIZE={} (RANK / WORLD_SIZE inherited from the environment, e.g. a SLURM step or '
                         'a parent launcher\'s shell, count as "already launched": unset them or pass the matching --gpus)'
                         .format(args.gpus, world))
    dist = None
    if D.dist_backend(torch.device('cuda')) == 'gloo' and launched:
        # device-less rendezvous (EMPOSE_DIST_BACKEND=gloo: the launcher's CPU test): join, meet the other ranks, then
        # go on to the device check below like any other run
        dist, rank, world = D.init_process_group(None, log=lambda m: print(m, file=sys.stderr, flush=True))
        dist.barrier()
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X; there is no CPU fallback')
    dev = torch.device('cuda', D.local_device_index(local_rank, 'bench.py'))
    torch.cuda.set_device(dev)
    for kv in args.option:
        from em_pose_amd import _lib
        name, _, value = kv.partition('=')
        _lib.check(_lib.lib().empose_set_option(name.encode(), int(value)))
    if args.workload == 'vertices':
        return run_vertices(args, dev)
    if dist is None and (launched and (world > 1 or args.force_dist)):
        dist, rank, world = D.init_process_group(dev)

    from em_pose_amd import _lib
    net, model = build_net(args.n_markers, not args.no_rnn, args.iterations)
    net = net.to(dev)
    B, F = args.batch, args.frames
    w, inputs = make_inputs(net, dev, B, F, seed=1000 + rank)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    out = None
    for _ in range(args.warmup):
        out = net.forward_tensors(*inputs)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = net.forward_tensors(*inputs)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    _lib.check(_lib.lib().empose_async_status())   # a cooperative LSTM kernel that gave up on a poll wrote NaN: not a result
    if dist is not None:
        cdev = D.collective_device(dev)     # the GPU under RCCL (the host under the gloo self-test)
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # the only data-path-adjacent collective: gather per-rank result checksums over RCCL
        mine = torch.stack([out['pose'].double().sum(), out['joints'].double().sum(),
                            torch.tensor(float(B * F), dtype=torch.float64, device=dev)]).to(cdev)
        allm = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allm, mine)
        frames_total = int(sum(m[2].item() for m in allm))
        assert all(torch.isfinite(m).all() for m in allm)
    else:
        frames_total = B * F
        assert torch.isfinite(out['pose']).all()
    barrier()
    # The same timed region once more on the fp32 MFMA instruction (option mlp_x3 = 0) when the headline ran the update
    # MLPs as three-piece bf16 products: reported beside the headline, never as `value`.
    lib0 = _lib.lib()
    x3_on = lib0.empose_get_option(b'mlp_x3') != 0 and net.config.m_hidden_size % 64 == 0 and B * F >= 64 * 128
    x3_options = {k: lib0.empose_get_option(k) for k in (b'mlp_x3', b'lstm_x3', b'rows_x3')}
    fp32_line = None
    if x3_on and not args.no_fp32_line:
        for k in x3_options:          # the whole path on the fp32 MFMA instruction: update MLPs, LSTM steps, heads, blend GEMMs
            _lib.check(lib0.empose_set_option(k, 0))
        for _ in range(max(args.warmup, 1)):
            out32 = net.forward_tensors(*inputs)
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            out32 = net.forward_tensors(*inputs)
        torch.cuda.synchronize()
        e32 = time.perf_counter() - t1
        for k, v in x3_options.items():     # back to what the headline ran with (a user's --option survives)
            _lib.check(lib0.empose_set_option(k, v))
        if dist is not None:
            t = torch.tensor([e32], dtype=torch.float64, device=D.collective_device(dev))
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            e32 = float(t.item())
        fp32_line = {'value': frames_total * args.steps / e32, 'ms_per_step': 1000.0 * e32 / args.steps,
                     'max_abs_diff_to_headline_outputs': float(max((out32[k] - out[k]).abs().max() for k in
                                                                   ('pose', 'shape', 'joints'))),
                     'what': 'options mlp_x3 = lstm_x3 = rows_x3 = 0: update MLPs (mlp_fused.hip), LSTM steps, init heads '
                             'and blend GEMMs on the fp32 MFMA instruction; same inputs, same timed region'}
        barrier()

    result = None
    if rank == 0:
        T = B * F
        value = frames_total * args.steps / elapsed
        fpf = flops_per_frame(net)
        result = {
            'metric': 'frames/sec LGD%s N=%d %d-sensor ws=%d; MPJPE vs ref (mm)' % ('-RNN' if net.rnn_init else '', net.N,
                                                                                   args.n_markers, F),
            'value': value, 'unit': 'frames/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1000.0 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': '%s: LGD%s %d-sensor, N=%d, ws=%d, batch %d windows per GPU, '
                                   '2x512 update MLPs, %s, synthetic SMPL-H-shaped body model (V=6890), '
                                   'random-init weights' % (baseline_config_label(net, args.n_markers, F, B),
                                                            '-RNN' if net.rnn_init else '', args.n_markers, net.N, F, B,
                                                            '2x512 LSTM init' if net.rnn_init else 'MLP init'),
                       'windows_per_gpu': B, 'frames_per_window': F, 'parallelism': 'window-sharded x%d' % world,
                       'process_group': None if dist is None else dist.get_backend(),
                       'keep_history': False,   # forward_tensors(); the five history tensors (N+1 entries each, 190 MB
                                                # per step at this batch) are written only when a caller asks for them

                       'dense_mflop_per_frame': fpf / 1e6,
                       'whole_path_tflops': value * fpf / 1e12,
                       'arithmetic': arithmetic_label(x3_options, x3_on, net.rnn_init)},
        }
        if fp32_line is not None:
            result['fp32_mfma_path'] = fp32_line
    if rank == 0 and not args.no_profile:
        lib = _lib.lib()
        lib.empose_profile_enable(1)
        psteps = 3
        for _ in range(psteps):
            net.forward_tensors(*inputs)
        prof = _lib.profile_read()
        lib.empose_profile_enable(0)
        total = sum(v[0] for v in prof.values())
        h = net.config.m_hidden_size
        if 'mlp_fused' in prof:
            # dominant kernel: both update nets, all their layers, one launch (csrc/mlp_fused.hip)
            ms, cnt = prof['mlp_fused']
            flops = 0.0
            for mlp in (net.pose_net_iter, net.shape_net_iter):
                flops += sum(2.0 * (B * F) * lin.in_features * lin.out_features for lin, _, _ in mlp.dense_specs())
            kname = 'mlp_fused_x3_kernel' if x3_on else 'mlp_fused_kernel'
            what = ' (the two update MLPs, 6 dense layers each, one launch per LGD iteration)'
        else:
            ms, cnt = prof['mlp_hidden_gemm']
            flops = 2 * 2.0 * (B * F) * h * h  # both update nets in one launch
            kname = lib.empose_profile_gemm_kernel_name(B * F, h, h, 2, 1).decode()
            what = ' (update-net hidden layer, both nets per launch)'
        avg_ms = ms / cnt
        timing = {'avg_launch_ms_all_launches_bracketed': avg_ms}
        if 'mlp_fused' in prof:
            # The pass above puts an event between ALL launches of the step (for the breakdown), which inflates every
            # interval by the cost of the event packets.  The dominant kernel is therefore timed again INSIDE the step, over
            # as many steps as the timed region ran, with events around its launches only.  `avg_launch_ms` is the plain
            # average of those intervals -- launch gap and event cost included, nothing subtracted: an upper bound of the
            # kernel's duration, the figure `rocprofv3 --kernel-trace --stats` reports as its average for the same command
            # (profiles/) agrees with it from below.  The cost of an empty event pair is reported beside it.
            _lib.check(lib.empose_profile_enable_only(b'mlp_fused'))
            for _ in range(max(args.steps, psteps)):
                net.forward_tensors(*inputs)
            solo = _lib.profile_read()
            lib.empose_profile_enable(0)
            avg_ms = solo['mlp_fused'][0] / solo['mlp_fused'][1]
            pair = solo['event_pair'][0] / solo['event_pair'][1] if 'event_pair' in solo else 0.0
            timing.update({'avg_launch_ms_bracketed_alone_in_step': avg_ms, 'launches_averaged': solo['mlp_fused'][1],
                           'empty_event_pair_ms': pair})
        ach = flops / (avg_ms * 1e-3) / 1e12
        traffic, traffic_src = (None, 'skipped (--no_traffic)')
        if world == 1 and not args.no_traffic:   # hardware counters: two child passes of this command under rocprofv3
            tail = ['--batch', str(B), '--frames', str(F), '--n_markers', str(args.n_markers),
                    '--iterations', str(args.iterations)] + (['--no_rnn'] if args.no_rnn else []) + \
                   [a for kv in args.option for a in ('--option', kv)]
            traffic, traffic_src = pmc_traffic_live(tail, kname)
        if traffic is None:
            why = traffic_src
            traffic, traffic_src = pmc_traffic(B * F, h, kname)
            if traffic_src:
                traffic_src += (' (rocprofv3 --pmc passes of this command in an EARLIER run, not measured by this '
                                'process: %s)' % why)
        # The roof of the kernel in ALGORITHMIC flops: the fp32 MFMA peak for the fp32 instruction; for the three-piece bf16
        # kernel the bf16 dense peak divided by the six piece products it executes per fp32 product (frac is then also
        # executed bf16 flops / bf16 peak).
        x3_kernel = kname == 'mlp_fused_x3_kernel'
        peak = PEAK_BF16_MFMA_TFLOPS / X3_PRODUCTS if x3_kernel else PEAK_FP32_MFMA_TFLOPS
        result['roofline'] = {'bound': 'mfma', 'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s',
                              'frac': ach / peak, 'traffic': traffic,
                              'peak_derivation': ('bf16 dense MFMA peak %.0f TFLOP/s / %d bf16 piece products per fp32 product'
                                                  % (PEAK_BF16_MFMA_TFLOPS, X3_PRODUCTS)) if x3_kernel else
                                                 'fp32 MFMA peak (v_mfma_f32_32x32x2_f32)',
                              'executed_matrix_tflops': ach * (X3_PRODUCTS if x3_kernel else 1),
                              'achieved_over_fp32_mfma_peak': ach / PEAK_FP32_MFMA_TFLOPS,
                              'traffic_source': traffic_src,
                              'traffic_raw_counters': getattr(pmc_traffic_live, 'raw', None),
                              'kernel': kname + what,
                              'avg_launch_ms': avg_ms, 'timing': timing, 'launches_per_step': cnt / psteps,
                              'flops_per_launch': flops,
                              'hbm_frac_on_algorithmic_bytes': value * 1162.0 / 1e9 / PEAK_HBM_GBS}
        result['breakdown_ms_per_step'] = {k: v[0] / psteps for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}
        result['breakdown_ms_per_step']['sum_of_kernels'] = total / psteps
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result['cpu_baseline'] = cpu_baseline(net, model, w, hip_out=out)
    if rank == 0 and world == 1 and not args.no_secondary and not args.no_rnn and (B, F, args.n_markers) == (1024, 32, 12):
        # (only beside the headline configuration; `value` / `metric` / `config` above are untouched by it)
        del inputs, out
        torch.cuda.empty_cache()
        result['secondary'] = secondary_workloads(args, dev, model)
    if rank == 0:
        print(json.dumps(result), file=results_out, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
