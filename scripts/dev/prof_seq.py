"""Dev: where the one-recording-at-a-time driver spends its time (B = 1, 256-frame chunks): per-kernel HIP-event totals
of one chunk and the wall time of the forward call."""
import sys, time
sys.path.insert(0, '.')
import torch
from em_pose_amd import _lib, synthetic
from em_pose_amd.bodymodels.smpl import SMPLLayer
from em_pose_amd.helpers.configuration import lgd_config
from em_pose_amd.nn.models import create_model
dev = torch.device('cuda:0')
net = create_model(lgd_config(12, True, 4), SMPLLayer(synthetic.make_model())).to(dev).eval()
B, F = 1, 256
w = synthetic.make_windows(B, F, 3, None) if False else None
g = torch.Generator().manual_seed(0)
mp, mo = torch.randn(B, F, 36, generator=g).to(dev), torch.randn(B, F, 108, generator=g).to(dev)
ot, orr = (torch.randn(B, 12, 3, generator=g) * 0.02).to(dev), torch.eye(3).expand(B, 12, 3, 3).contiguous().to(dev)
for _ in range(3): net.forward_tensors(mp, mo, ot, orr)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(10): net.forward_tensors(mp, mo, ot, orr)
torch.cuda.synchronize()
print('forward B=1 F=256: %.2f ms wall per call' % ((time.time() - t0) / 10 * 1e3))
t0 = time.time()
for _ in range(10): net.forward_tensors(mp, mo, ot, orr)
t_launch = (time.time() - t0) / 10 * 1e3
torch.cuda.synchronize()
print('  of which host-side launch time (no sync): %.2f ms' % t_launch)
lib = _lib.lib(); lib.empose_profile_enable(1)
for _ in range(3): net.forward_tensors(mp, mo, ot, orr)
p = _lib.profile_read(); lib.empose_profile_enable(0)
print({k: (round(v[0] / 3, 3), v[1] // 3) for k, v in sorted(p.items(), key=lambda kv: -kv[1][0])})
