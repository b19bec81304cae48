"""Dev: time the SMPL evaluation kernels alone at T=32768 (fwd+bwd), optionally stopping the chain kernel early."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from em_pose_amd import _lib, synthetic
from em_pose_amd.bodymodels.smpl import SMPLLayer
from em_pose_amd.helpers.configuration import lgd_config
from em_pose_amd.nn.models import create_model
dev = torch.device('cuda:0')
model = synthetic.make_model()
net = create_model(lgd_config(12, False, 1, hidden=32), SMPLLayer(model)).to(dev).eval()
h = net._ensure_handle(dev)
lib = _lib.lib()
T, F = int(os.environ.get('T', 32768)), 32
g = torch.Generator(device='cpu').manual_seed(0)
th = (torch.randn(T, 66, generator=g) * 0.25).to(dev); be = torch.randn(T, 10, generator=g).to(dev)
o_r = torch.eye(3).expand(T // F, 12, 3, 3).contiguous().to(dev); o_t = (torch.randn(T // F, 12, 3, generator=g) * 0.02).to(dev)
tgt = torch.randn(T, 144, generator=g).to(dev); sc = torch.ones(T, device=dev)
pos, ori, jo = (torch.empty(T, n, device=dev) for n in (36, 108, 66))
gt, gb = torch.empty(T, 66, device=dev), torch.empty(T, 10, device=dev)
nb = lib.empose_smpl_workspace_bytes(h, T); ws = torch.empty(nb, dtype=torch.uint8, device=dev)
def run():
    _lib.check(lib.empose_smpl_sensors_fwd_bwd(h, T, F, _lib.dptr(th), 66, _lib.dptr(be), 10, _lib.dptr(o_r), _lib.dptr(o_t),
               _lib.dptr(tgt), 144, _lib.dptr(sc), _lib.dptr(pos), _lib.dptr(ori), _lib.dptr(jo), _lib.dptr(gt), 66, _lib.dptr(gb), 10,
               _lib.dptr(ws), nb, None))
for _ in range(3): run()
torch.cuda.synchronize()
lib.empose_profile_enable(1)
for _ in range(10): run()
p = _lib.profile_read(); lib.empose_profile_enable(0)
print('stop', os.environ.get('EMPOSE_CHAIN_STOP', '0'), {k: round(v[0] / v[1] * 1000, 1) for k, v in p.items()}, 'us/launch')
