"""Dev: the headline-shaped forward + the full-mesh evaluation repeated beside a stream of element-wise kernels: how many
values differ from the run alone on the device?  (EMPOSE_LIB_PATH selects a lab build, scripts/dev/x3_shared_simd_lab.sh.)"""
import os, sys; sys.path.insert(0, '.')
import torch
from em_pose_amd import _lib, synthetic
if os.environ.get('EMPOSE_LIB_PATH'):
    _lib.LIB_PATH = os.environ['EMPOSE_LIB_PATH']
from em_pose_amd.bodymodels.smpl import SMPLLayer
from em_pose_amd.helpers.configuration import lgd_config
from em_pose_amd.nn.models import create_model
DEV = 'cuda:0'
model = synthetic.make_model()
torch.manual_seed(3)
net = create_model(lgd_config(12, True, 4), SMPLLayer(model)).eval().to(DEV)
B, F = 512, 32
g = torch.Generator().manual_seed(7)
inputs = [torch.randn(B, F, 36, generator=g).to(DEV), torch.randn(B, F, 108, generator=g).to(DEV),
          (0.02 * torch.randn(B, 12, 3, generator=g)).to(DEV), torch.eye(3).repeat(B, 12, 1, 1).to(DEV)]
smpl = SMPLLayer(model).to(DEV)
T = 4096
kw = dict(poses_body=(torch.randn(T, 63, generator=g) * 0.5).to(DEV), betas=torch.randn(T, 10, generator=g).to(DEV),
          poses_root=(torch.randn(T, 3, generator=g) * 0.5).to(DEV))
def run():
    res = net.forward_tensors(*inputs)
    v, j = smpl(**kw)
    torch.cuda.synchronize()
    return [res[k].clone() for k in ('pose', 'shape', 'joints')] + [v.clone()]
alone = run()
side = torch.cuda.Stream()
big = torch.randn(1 << 26, device=DEV)
for rep in range(6):
    with torch.cuda.stream(side):
        for _ in range(400):
            big.mul_(1.0000001).add_(1e-9)
    got = run()
    diffs = [(int((a != b).sum()), float((a - b).abs().max())) for a, b in zip(got, alone)]
    print('repetition %d beside the other stream: values that differ (count, max) in pose / shape / joints / vertices: %s' % (rep, diffs))
    side.synchronize()
