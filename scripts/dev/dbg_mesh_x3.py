"""Dev: repeatability / agreement of the full-mesh kernels (option mesh_x3 = 0 / 1 / 2) and where they differ."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
import os
from em_pose_amd import _lib, synthetic
if os.environ.get('EMPOSE_LIB_PATH'):
    _lib.LIB_PATH = os.environ['EMPOSE_LIB_PATH']   # dev: a lab build of the library
from em_pose_amd.bodymodels.smpl import SMPLLayer
dev = 'cuda:0'
lib = _lib.lib()
smpl = SMPLLayer(synthetic.make_model()).to(dev)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
g = torch.Generator().manual_seed(5)
kw = dict(poses_body=(torch.randn(T, 63, generator=g) * 0.5).to(dev), betas=torch.randn(T, 10, generator=g).to(dev),
          poses_root=(torch.randn(T, 3, generator=g) * 0.5).to(dev))
outs = {}
for opt in (0, 2, 1):
    _lib.check(lib.empose_set_option(b'mesh_x3', opt))
    runs = [smpl(**kw)[0].clone() for _ in range(4)]
    torch.cuda.synchronize()
    outs[opt] = runs[0]
    for k, r in enumerate(runs[1:]):
        d = (r - runs[0]).abs()
        nz = (d > 0).nonzero()
        print('opt', opt, 'run', k + 1, 'vs run 0: differing elements', nz.shape[0], 'max', float(d.max()))
        if nz.shape[0]:
            f, v, c = nz[:, 0].cpu().numpy(), nz[:, 1].cpu().numpy(), nz[:, 2].cpu().numpy()
            print('   frame%64', np.unique(f % 64)[:20], ' tiles', np.unique(v // 32)[:20], ' v%32', np.unique(v % 32)[:32], ' coord', np.unique(c),
                  ' tile%4', np.unique((v // 32) % 4))
for opt in (1, 2):
    d = (outs[opt] - outs[0]).abs()
    print('opt', opt, 'vs fp32 kernel: max', float(d.max()), 'elements > 1e-5:', int((d > 1e-5).sum()))
    big = (d > 1e-5).nonzero()
    if big.shape[0]:
        f, v, c = big[:, 0].cpu().numpy(), big[:, 1].cpu().numpy(), big[:, 2].cpu().numpy()
        print('   frame%64', np.unique(f % 64)[:20], ' tiles', np.unique(v // 32)[:20], ' v%32', np.unique(v % 32)[:32], ' coord', np.unique(c))
