"""Dev: the split-bf16 full-mesh kernel against the fp32 kernel on several frame counts, four launches each (count,
location and coordinate of every element off by more than 1e-4), then ten repetitions of 2048 and 16384 frames compared
bit for bit.  This is the harness that showed the two-waves-per-SIMD build of the kernel returning sporadically wrong
x coordinates (mesh.hip); optional argv[1]: path of an alternative libempose_hip.so build to load."""
import numpy as np, torch, sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import em_pose_amd._lib as L
if len(sys.argv) > 1:
    L.LIB_PATH = os.path.abspath(sys.argv[1])
from em_pose_amd.bodymodels.smpl import SMPLLayer
from em_pose_amd import synthetic
model = synthetic.make_model()
fast = SMPLLayer(model, arithmetic='bf16x3').to('cuda')
exact = SMPLLayer(model).to('cuda')
rng = np.random.default_rng(21)
tot = 0
for n, wt in ((64, False), (700, False), (2048, True)):
    pose = torch.from_numpy(rng.normal(0, 0.5, size=(n, 63)).astype(np.float32)).cuda()
    root = torch.from_numpy(rng.normal(0, 0.5, size=(n, 3)).astype(np.float32)).cuda()
    betas = torch.from_numpy(rng.normal(0, 1.5, size=(n, 16)).astype(np.float32)).cuda()
    trans = torch.from_numpy(rng.normal(0, 1, size=(n, 3)).astype(np.float32)).cuda() if wt else None
    v2, j2 = exact(poses_body=pose, betas=betas, poses_root=root, trans=trans)
    v2 = v2.clone()
    for rep in range(4):
        v, j = fast(poses_body=pose, betas=betas, poses_root=root, trans=trans)
        d = (v - v2).abs()
        bad = (d > 1e-4).nonzero()
        tot += bad.shape[0]
        print(n, wt, rep, float(d.max()), bad.shape[0], bad[:2].tolist(), bad[-2:].tolist(),
              'coords', sorted(set(bad[:, 2].tolist())), 'frames%64', sorted(set((bad[:, 0] % 64).tolist()))[:6],
              'v%32>=16', bool(((bad[:, 1] % 32) >= 16).all()) if bad.shape[0] else None)
print(sys.argv[1:], 'TOTAL BAD', tot)
# nondeterminism between repetitions (independent of the reference)
for n in (2048, 16384):
    pose = torch.from_numpy(rng.normal(0, 0.5, size=(n, 63)).astype(np.float32)).cuda()
    root = torch.from_numpy(rng.normal(0, 0.5, size=(n, 3)).astype(np.float32)).cuda()
    betas = torch.from_numpy(rng.normal(0, 1.5, size=(n, 16)).astype(np.float32)).cuda()
    ref = fast(poses_body=pose, betas=betas, poses_root=root)[0].clone()
    ex = exact(poses_body=pose, betas=betas, poses_root=root)[0]
    print(n, 'max |fast - exact|', float((ref - ex).abs().max()))
    del ex
    nbad = 0
    for rep in range(10):
        o = fast(poses_body=pose, betas=betas, poses_root=root)[0]
        bad = (o != ref).nonzero()
        nbad += bad.shape[0]
        if bad.shape[0]:
            print('  rep', rep, 'differs from rep 0 in', bad.shape[0], 'coords', sorted(set(bad[:, 2].tolist())),
                  'frames%64', sorted(set((bad[:, 0] % 64).tolist()))[:8])
        del o
    print(n, 'nondeterministic elements over 10 reps:', nbad)
