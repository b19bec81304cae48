"""Randomized LGD / LGD-RNN forwards (small test mesh) against the oracle: batch sizes across the three LSTM regimes,
ragged lengths, missing sensors, carried LSTM state, and the kernel variants behind the same call.

    python tests/fuzz/fuzz_lgd.py <seed> <seconds | n=CASES> [force=<tile>,<fuse>,<suppression>,<two parts>] [extra]

`run()` is what the command line and tests/test_fuzz_slice.py (a fixed-seed slice inside `pytest -m gpu`) both call.
The case sequence of a seed depends only on (seed, extra_nets): `start` skips the evaluation of the first cases but
makes their random draws, so case k of a seed is the same inputs whether it is reached by running or by skipping.
"""
import sys
import time

if __name__ == '__main__':
    sys.path.insert(0, '.')
    sys.path.insert(0, 'tests')

import numpy as np
import torch

GOLDEN_NETS = ('lgdrnn12_n4_carry', 'lgdrnn6_n2', 'lgd12_n4')
# random-init nets of shapes the golden ones do not have: LSTM narrower than the heads' staging tile, LSTM wider than the
# update nets, hidden widths below and above the input width, one and three LSTM-free / LSTM iterations
EXTRA_NETS = (('x_rnn16_h64_m12_n2', dict(n_markers=12, rnn=True, N=2, hidden=64, rnn_hidden=16)),
              ('x_rnn64_h16_m6_n3', dict(n_markers=6, rnn=True, N=3, hidden=16, rnn_hidden=64)),
              ('x_rnn8_h128_m12_n1', dict(n_markers=12, rnn=True, N=1, hidden=128, rnn_hidden=8)),
              ('x_mlp_h48_m6_n2', dict(n_markers=6, rnn=False, N=2, hidden=48, rnn_hidden=32)))
BATCHES = (1, 2, 4, 7, 16, 17, 33, 64, 130, 257)
BATCHES_SLICE = (1, 2, 7, 16, 17, 33, 64, 256, 257)


def build_nets(dev, extra_nets=False):
    try:
        import helpers as H
    except ImportError:
        from tests import helpers as H
    from em_pose_amd.bodymodels.smpl import SMPLLayer
    from em_pose_amd.helpers.configuration import lgd_config
    from em_pose_amd.nn.models import create_model
    from oracle import torch_ref as R
    model = H.small_model()
    bm = R.BodyModelTensors(model)
    nets = {}
    for name in GOLDEN_NETS:
        case = H.load_case(name)
        meta = case['meta']
        vids = [int(v) for v in meta['vertex_ids']]
        cfg = lgd_config(int(meta['n_markers']), bool(meta['rnn']), int(meta['N']), hidden=32, rnn_hidden=32)
        net = create_model(cfg, SMPLLayer(model))
        net.load_state_dict(H.sd_to_torch(case['sd']), strict=False)
        net.vertex_ids = vids
        nets[name] = (net.to(dev).eval(), H.sd_to_torch(case['sd']),
                      dict(n_markers=int(meta['n_markers']), rnn=bool(meta['rnn']), N=int(meta['N']), rnn_hidden=32),
                      vids, R.sensor_tables(model['f'], vids))
    if extra_nets:
        vids = nets[GOLDEN_NETS[0]][3]
        for i, (name, kw) in enumerate(EXTRA_NETS):
            gen_state = torch.random.get_rng_state()
            torch.manual_seed(77000 + i)
            cfg = lgd_config(kw['n_markers'], kw['rnn'], kw['N'], hidden=kw['hidden'], rnn_hidden=kw['rnn_hidden'])
            net = create_model(cfg, SMPLLayer(model)).eval()
            g = torch.Generator().manual_seed(78000 + i)
            with torch.no_grad():
                for m in net.modules():
                    if isinstance(m, torch.nn.BatchNorm1d):
                        m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                        m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
            torch.random.set_rng_state(gen_state)
            net.vertex_ids = vids
            sd = {k: v.detach().clone() for k, v in net.state_dict().items() if not k.startswith('smpl.')}
            nets[name] = (net.to(dev).eval(), sd, kw, vids, R.sensor_tables(model['f'], vids))
    return model, bm, nets


N_FP_SAMPLES, N_FP_PROJ = 512, 4
FP_KEYS = ('pose_hat', 'root_ori_hat', 'shape_hat', 'joints_hat', 'rnn_h')


def fingerprint(case, outputs):
    """What the fixture of a fixed-seed slice keeps of the ORACLE's outputs of one case (the outputs themselves would be
    tens of MB): per output -- the valid frames of pose / root orientation / shape / joints, the final LSTM state -- 512
    entries at seeded positions and 4 seeded Gaussian projections divided by sqrt(n) (an error of e in every entry moves
    such a projection by about e).  `outputs`: {key: 1-D float array}."""
    fp = {}
    for j, k in enumerate(FP_KEYS):
        if k not in outputs:
            continue
        x = np.asarray(outputs[k], dtype=np.float64).reshape(-1)
        rng = np.random.default_rng(900000 + 10 * case + j)
        idx = rng.integers(0, x.size, size=min(N_FP_SAMPLES, x.size))
        z = rng.standard_normal((N_FP_PROJ, x.size))
        fp[k + '/samples'] = x[idx].astype(np.float32)
        fp[k + '/proj'] = (z @ x) / np.sqrt(x.size)
    return fp


def load_fixture(path):
    z = np.load(path)
    cache = {}
    for key in z.files:
        case, rest = key.split('/', 1)
        cache.setdefault(int(case[4:]), {})[rest] = z[key]
    return cache


def run(seed=0, seconds=None, n_cases=None, force=None, extra_nets=False, start=0, batches=BATCHES, dev='cuda:0',
        log=print, tol=1e-4, oracle_cache=None, record=None):
    """Returns {'n', 'worst', 'worst_case', 'above_1e5': [(case, err, description)], 'variants', 'errors': [per case]}.
    Raises AssertionError on the first case at or above `tol`.
    `record` (a dict): CPU only -- no HIP call; the oracle's fingerprint of every case goes into it (the fixture of a
    fixed-seed slice, generated in the build container).  `oracle_cache` (such a fixture, `load_fixture`): cases it holds
    are compared with the recorded fingerprints instead of running the oracle on the GPU box's host (0.9 s per case, the
    cost that kept the in-suite slice at 48 cases) -- sampled entries and projections instead of every entry; a case
    above 1e-5 is still re-run through the oracle in full for the conditioning analysis."""
    from em_pose_amd import _lib, synthetic
    from oracle import torch_ref as R
    try:
        import helpers as H
    except ImportError:
        from tests import helpers as H
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)   # the carried LSTM states come from torch's generator: same seed, same cases
    model, bm, nets = build_nets(dev, extra_nets)
    t_end = time.time() + (seconds if seconds is not None else 1e9)
    n, worst, worst_case, above, variants, errors = 0, 0.0, None, [], {}, []
    side = torch.cuda.Stream() if record is None else None
    lib = _lib.lib()
    try:
        while time.time() < t_end and (n_cases is None or n < n_cases):
            name = list(nets)[int(rng.integers(0, len(nets)))]
            net, sd, meta, vids, tables = nets[name]
            B = int(rng.choice(list(batches)))
            F = int(rng.integers(1, 20))
            skip = n < start
            lens = rng.integers(1, F + 1, size=B)
            lens[0] = F
            masks = None
            if rng.integers(0, 2):
                masks = (rng.uniform(size=(B, F, 12)) > 0.05).astype(np.float32)
            rnn, Hr = bool(meta['rnn']), int(meta['rnn_hidden'])
            state = None
            if rnn and rng.integers(0, 2):
                state = (0.3 * torch.randn(2, B, Hr), 0.3 * torch.randn(2, B, Hr))
            # the kernel variants behind the same call -- frame-per-lane SMPL kernels forced on / off, the small kernels
            # inside the blend GEMMs or on their own, missing-sensor replacement in the packing kernel (raw readings with
            # garbage under the missing sensors), the forward in two parts on two streams
            tile, fuse = int(rng.choice([0, 2])), int(rng.integers(0, 2))
            want_supp = masks is not None and bool(rng.integers(0, 2))
            two_parts = bool(rng.integers(0, 2))
            if force:   # replay with given variants (the random draws above are still made: same case sequence)
                tile, fuse = force[0], force[1]
                want_supp = masks is not None and bool(force[2])
                two_parts = bool(force[3])
            if skip:
                n += 1
                continue

            def sensors(poses, betas, o_r, o_t):
                with torch.no_grad():
                    p, o, _ = R.estimated_markers(bm, tables, vids, torch.from_numpy(poses), torch.from_numpy(betas),
                                                  torch.from_numpy(o_r), torch.from_numpy(o_t))
                return p.numpy(), o.numpy()
            w = synthetic.make_windows(B, F, 1000 + n, sensors)
            if masks is not None:
                w['marker_masks'] = masks
            inp = H.oracle_inputs(w, sl=lens)
            valid = (torch.arange(F)[None, :] < torch.as_tensor(lens)[:, None]).numpy()
            cached = oracle_cache.get(n) if oracle_cache is not None else None

            def oracle32():
                return R.ief_forward(sd, bm, tables, vids, inp, n_markers=int(meta['n_markers']), N=int(meta['N']),
                                     rnn_init=rnn, rnn_state=state)

            def flat(pose_hat, root, shape, joints, h):
                out = {'pose_hat': pose_hat[valid], 'root_ori_hat': root[valid], 'shape_hat': shape[valid],
                       'joints_hat': joints[valid]}
                if h is not None:
                    out['rnn_h'] = h
                return out
            want = tr = None
            if cached is None:
                want, tr = oracle32()
            if record is not None:
                record[n] = fingerprint(n, flat(want['pose_hat'].numpy(), want['root_ori_hat'].numpy(),
                                                want['shape_hat'].numpy(), want['joints_hat'].numpy(),
                                                tr['rnn_state'][0].numpy() if rnn else None))
                n += 1
                continue
            g = lambda t: None if t is None else t.to(dev)
            _lib.check(lib.empose_set_option(b'smpl_tile', tile))
            _lib.check(lib.empose_set_option(b'smpl_fuse', fuse))
            mp, mo, kw = inp['marker_pos'], inp['marker_oris'], {}
            if want_supp:
                miss = (inp['marker_masks'] != 1).reshape(B, F, 12, 1)
                mp = torch.where(miss.expand(B, F, 12, 3).reshape(B, F, 36), torch.full_like(mp, 9.0), mp)
                mo = torch.where(miss.expand(B, F, 12, 9).reshape(B, F, 108), torch.full_like(mo, -2.0), mo)
                kw['suppress_mask_value'] = 0.0
            net.iter_stream = side if two_parts else None
            res = net.forward_tensors(g(mp), g(mo), g(inp['offset_t']), g(inp['offset_r']),
                                      marker_masks=g(inp['marker_masks']), seq_lengths=g(inp['seq_lengths']),
                                      state=None if state is None else tuple(g(t) for t in state), **kw)
            if two_parts:
                torch.cuda.current_stream().wait_event(net.outputs_ready)
            net.iter_stream = None
            vkey = (tile, fuse, 'suppress_mask_value' in kw, two_parts)
            variants[vkey] = variants.get(vkey, 0) + 1
            err = 0.0
            pose_np = res['pose'].cpu().numpy()
            if cached is not None:
                mine = fingerprint(n, flat(pose_np[:, :, 3:], pose_np[:, :, :3], res['shape'].cpu().numpy(),
                                           res['joints'].cpu().numpy(), res['state'][0].cpu().numpy() if rnn else None))
                assert set(mine) == set(cached), (sorted(mine), sorted(cached))
                for k, v in mine.items():
                    assert v.shape == cached[k].shape, (n, k, v.shape, cached[k].shape)   # same case, same sizes
                    err = max(err, float(np.abs(v.astype(np.float64) - cached[k]).max()))
            else:
                for got, ref in ((pose_np[:, :, 3:], want['pose_hat'].numpy()),
                                 (pose_np[:, :, :3], want['root_ori_hat'].numpy()),
                                 (res['shape'].cpu().numpy(), want['shape_hat'].numpy()),
                                 (res['joints'].cpu().numpy(), want['joints_hat'].numpy())):
                    err = max(err, float(np.abs(got - ref)[valid].max()))
                if rnn:
                    err = max(err, float((res['state'][0].cpu() - tr['rnn_state'][0]).abs().max()))
            desc = (name, dict(B=B, F=F, masks=masks is not None, state=state is not None), vkey)
            errors.append(err)
            if err > 1e-5 or not np.isfinite(err):
                if want is None:
                    want, tr = oracle32()
                # How much of that is the input's conditioning?  The same case through the oracle in float64: the distance
                # of the fp32 ORACLE from it is what fp32 arithmetic costs on this input whoever does it.
                sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
                want64, tr64 = R.ief_forward(sd64, R.BodyModelTensors(model, dtype=torch.float64), tables, vids,
                                             H.oracle_inputs(w, sl=lens, dtype=torch.float64),
                                             n_markers=int(meta['n_markers']), N=int(meta['N']), rnn_init=rnn,
                                             rnn_state=None if state is None else tuple(t.double() for t in state))
                e_hip, e_o32 = 0.0, 0.0
                for got, k, sl_ in ((res['pose'], 'pose_hat', slice(3, None)), (res['pose'], 'root_ori_hat', slice(0, 3)),
                                    (res['shape'], 'shape_hat', slice(None)), (res['joints'], 'joints_hat', slice(None))):
                    ref64 = want64[k].numpy()
                    e_hip = max(e_hip, float(np.abs(got.cpu().numpy()[:, :, sl_] - ref64)[valid].max()))
                    e_o32 = max(e_o32, float(np.abs(want[k].numpy() - ref64)[valid].max()))
                # ... and the case's own sensitivity, free of any implementation's rounding luck: the float64 oracle with
                # the body-model constants (template, blend shapes, regressor, skin weights) and the inputs moved by one
                # fp32 unit in the last place (random signs, two draws) -- the noise any fp32 evaluation of the vertices
                # commits -- and how far the outputs move
                sens = 0.0
                prng = np.random.default_rng(12345 + n)
                ulp = lambda a: a.astype(np.float64) * (1.0 + 6e-8 * np.sign(prng.standard_normal(a.shape)))
                for draw in range(2):
                    wp = dict(w)
                    for k in ('marker_pos', 'marker_oris', 'offset_t'):
                        wp[k] = ulp(w[k])
                    model_p = {k: (ulp(v) if v.dtype.kind == 'f' else v) for k, v in model.items()}
                    want_p, _ = R.ief_forward(sd64, R.BodyModelTensors(model_p, dtype=torch.float64), tables, vids,
                                              H.oracle_inputs(wp, sl=lens, dtype=torch.float64),
                                              n_markers=int(meta['n_markers']), N=int(meta['N']), rnn_init=rnn,
                                              rnn_state=None if state is None else tuple(t.double() for t in state))
                    for k in ('pose_hat', 'root_ori_hat', 'shape_hat', 'joints_hat'):
                        sens = max(sens, float(np.abs(want_p[k].numpy() - want64[k].numpy())[valid].max()))
                above.append((n, err, desc, {'hip_vs_f64': e_hip, 'oracle_f32_vs_f64': e_o32, 'one_ulp_sensitivity': sens}))
                log('case %d above 1e-5: %.3e %s; vs the float64 oracle: HIP %.3e, fp32 oracle %.3e; the float64 outputs '
                    'move by %.3e when body-model constants and inputs move by one fp32 ulp' % (n, err, desc, e_hip, e_o32, sens))
            if err > worst or not np.isfinite(err):
                worst_case = (n,) + desc
            worst = max(worst, err) if np.isfinite(err) else float('nan')
            n += 1
            assert err < tol, 'LGD MISMATCH seed %d case %d %s: %r' % (seed, n - 1, desc, err)
    finally:
        if record is None:
            _lib.check(lib.empose_set_option(b'smpl_tile', 1))
            _lib.check(lib.empose_set_option(b'smpl_fuse', 1))
        for net in nets.values():
            net[0].iter_stream = None
    return {'n': n - start, 'worst': worst, 'worst_case': worst_case, 'above_1e5': above, 'variants': variants,
            'errors': errors}


SLICE_SEED, SLICE_CASES = 4101, 120     # the in-suite slice (tests/test_fuzz_slice.py) and its fixture


def slice_fixture_path():
    import os
    return os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'golden',
                        'fuzz_lgd_slice_%d.npz' % SLICE_SEED)


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'record':
    # build container, CPU only: the oracle's fingerprints of the in-suite slice -> tests/golden/fuzz_lgd_slice_<seed>.npz
    rec = {}
    run(seed=SLICE_SEED, n_cases=SLICE_CASES, extra_nets=True, batches=BATCHES_SLICE, dev='cpu', record=rec)
    np.savez_compressed(slice_fixture_path(), **{'case%d/%s' % (c, k): v for c, fp in rec.items() for k, v in fp.items()})
    print('wrote', slice_fixture_path(), len(rec), 'cases')
    sys.exit(0)

if __name__ == '__main__':
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    arg = sys.argv[2] if len(sys.argv) > 2 else '60'
    rest = sys.argv[3:]
    force = next(([int(v) for v in a[6:].split(',')] for a in rest if a.startswith('force=')), None)
    r = run(seed, n_cases=int(arg[2:]) if arg.startswith('n=') else None,
            seconds=None if arg.startswith('n=') else float(arg), force=force, extra_nets='extra' in rest)
    print('lgd: %d random cases, worst abs error %.2e' % (r['n'], r['worst']), 'at', r['worst_case'])
    print('     (smpl_tile, smpl_fuse, on-device suppression, two-part forward) -> cases:', sorted(r['variants'].items()))
