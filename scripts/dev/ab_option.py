"""Dev: A/B of one kernel-variant option on the headline forward inside ONE process (boxes differ by a few percent):
python scripts/dev/ab_option.py <option> [batch] -- alternates option = 1 / 0 in blocks of 10 forwards, five rounds."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from em_pose_amd import _lib, synthetic
from em_pose_amd.bodymodels.smpl import SMPLLayer
from em_pose_amd.helpers.configuration import lgd_config
from em_pose_amd.nn.models import create_model

opt = sys.argv[1].encode()
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device('cuda:0')
torch.manual_seed(0)
net = create_model(lgd_config(12, True, 4), SMPLLayer(synthetic.make_model())).to(dev).eval()
g = torch.Generator().manual_seed(1)
args = [torch.randn(B, 32, 36, generator=g).to(dev), torch.randn(B, 32, 108, generator=g).to(dev),
        (0.02 * torch.randn(B, 12, 3, generator=g)).to(dev), torch.eye(3).expand(B, 12, 3, 3).contiguous().to(dev)]
lib = _lib.lib()
for _ in range(3):
    net.forward_tensors(*args)
torch.cuda.synchronize()
res = {0: [], 1: []}
for rnd in range(5):
    for mode in (1, 0):
        _lib.check(lib.empose_set_option(opt, mode))
        net.forward_tensors(*args)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            net.forward_tensors(*args)
        torch.cuda.synchronize()
        res[mode].append((time.perf_counter() - t0) / 10 * 1e3)
_lib.check(lib.empose_set_option(opt, 1))
for mode in (1, 0):
    print('%s = %d: ms per forward %s  median %.3f' % (opt.decode(), mode, ['%.3f' % v for v in res[mode]], float(np.median(res[mode]))))
