"""Random shapes through empose_linear_f32_ex (bias, PReLU, residual) against fp64 NumPy, and random frame counts through
the full-mesh kernel against the oracle on the small test mesh.

    python tests/fuzz/fuzz_linear_mesh.py <seed> <seconds>

`run_linear()` / `run_mesh()` are shared with tests/test_fuzz_slice.py (a fixed-seed slice inside `pytest -m gpu`)."""
import sys
import time

if __name__ == '__main__':
    sys.path.insert(0, '.')
    sys.path.insert(0, 'tests')

import numpy as np
import torch


def run_linear(seed=0, seconds=None, n_cases=None, log=print, dev='cuda:0'):
    from em_pose_amd import _lib
    rng = np.random.default_rng(seed)
    lib = _lib.lib()
    t_end = time.time() + (seconds if seconds is not None else 1e9)
    n, worst, worst_case = 0, 0.0, None
    while time.time() < t_end and (n_cases is None or n < n_cases):
        M = int(rng.choice([1, 2, 31, 64, 65, 127, 128, 200, 513, 1000, 4096, 24576 + int(rng.integers(0, 300))]))
        N = int(rng.choice([1, 3, 10, 32, 66, 100, 128, 200, 256, 300, 320, 512, 520]))
        K = int(rng.integers(1, 150)) * 4
        g = torch.Generator().manual_seed(int(rng.integers(0, 1 << 30)))
        x = torch.randn(M, K, generator=g)
        w = torch.randn(N, K, generator=g) / np.sqrt(K)
        b = torch.randn(N, generator=g)
        act = int(rng.integers(0, 2))
        slope = float(rng.uniform(-0.5, 1.5))
        use_res = bool(rng.integers(0, 2))
        res = torch.randn(M, N, generator=g) if use_res else None
        want = x.double() @ w.double().t() + b.double()
        if act:
            want = torch.where(want >= 0, want, slope * want)
        if use_res:
            want = want + res.double()
        xg, wg, bg = x.to(dev), w.to(dev), b.to(dev)
        rg = res.to(dev) if use_res else None
        out = torch.empty(M, N, device=dev)
        _lib.check(lib.empose_linear_f32_ex(_lib.dptr(xg), K, _lib.dptr(wg), K, _lib.dptr(out), N, M, N, K, None,
                                            _lib.dptr(bg), _lib.dptr(rg), N if use_res else 0, act, slope, None))
        torch.cuda.synchronize()
        err = float((out.cpu().double() - want).abs().max())
        tol = 2e-5 * max(1.0, float(want.abs().max()))
        desc = dict(M=M, N=N, K=K, act=act, slope=slope, res=use_res)
        if err > worst:
            worst_case = (n, desc)
        worst = max(worst, err) if np.isfinite(err) else float('nan')
        n += 1
        assert err < tol, 'LINEAR MISMATCH seed %d case %d %s: %r (tol %r)' % (seed, n - 1, desc, err, tol)
    return {'n': n, 'worst': worst, 'worst_case': worst_case}


def run_mesh(seed=0, seconds=None, n_cases=None, log=print, dev='cuda:0', tol=3e-5):
    try:
        import helpers as H
    except ImportError:
        from tests import helpers as H
    from em_pose_amd.bodymodels.smpl import SMPLLayer
    from oracle import torch_ref as R
    rng = np.random.default_rng(seed)
    model = H.small_model()
    layers = {}
    for conv in ('smplx', 'so3'):
        bm = R.BodyModelTensors(model, rodrigues_convention=conv)
        for arith in ('f32', 'bf16x3'):
            layers[(conv, arith)] = (bm, SMPLLayer(model, rodrigues_convention=conv, arithmetic=arith).to(dev))
    t_end = time.time() + (seconds if seconds is not None else 1e9)
    n, worst = 0, {'f32': 0.0, 'bf16x3': 0.0}
    while time.time() < t_end and (n_cases is None or n < n_cases):
        T = int(rng.choice([1, 2, 63, 64, 65, 128, 200, 1000, 5000]))
        g = torch.Generator().manual_seed(n)
        scale = float(rng.choice([1e-3, 0.4, 1.5]))   # incl. angles near the guard of the axis-angle map
        pose, root = scale * torch.randn(T, 63, generator=g), 0.5 * torch.randn(T, 3, generator=g)
        betas, trans = torch.randn(T, 10, generator=g), torch.randn(T, 3, generator=g)
        use_tr = bool(rng.integers(0, 2))
        conv = ('smplx', 'so3')[int(rng.integers(0, 2))]
        # round 6: the default ('f32') path is the three-piece bf16 kernel (mesh_x3.hip) in one of its three forms, or the
        # fp32 MFMA kernel (option mesh_x3 = 0)
        from em_pose_amd import _lib
        _lib.check(_lib.lib().empose_set_option(b'mesh_x3', int(rng.integers(0, 4))))
        for arith in ('f32', 'bf16x3'):
            bm, smpl = layers[(conv, arith)]
            v_ref, j_ref = R.smpl_fk(bm, pose, betas, root, trans if use_tr else None)
            v, j = smpl(poses_body=pose.to(dev), betas=betas.to(dev), poses_root=root.to(dev),
                        trans=trans.to(dev) if use_tr else None)
            assert tuple(j.shape) == (T, 52, 3)
            err = max(float((v.cpu() - v_ref).abs().max()), float((j.cpu() - j_ref).abs().max()))
            worst[arith] = max(worst[arith], err) if np.isfinite(err) else float('nan')
            assert err < tol, 'MESH MISMATCH seed %d case %d %s: %r' % (seed, n, (T, use_tr, conv, arith, scale), err)
        n += 1
    return {'n': n, 'worst': worst}


if __name__ == '__main__':
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
    r = run_linear(seed, seconds=budget)
    print('linear: %d random cases, worst abs error %.2e at %s' % (r['n'], r['worst'], r['worst_case']))
    r = run_mesh(seed, seconds=budget / 2)
    print('mesh: %d random cases x (f32, bf16x3), both Rodrigues conventions, worst abs error %.2e / %.2e'
          % (r['n'], r['worst']['f32'], r['worst']['bf16x3']))
