"""Dev: the stand-alone LSTM entry point (2 x 512, input 60) on medium batches: us per wavefront step
(B <= 256: lstm_mid_kernel, above: lstm_chain_kernel; the threshold constant LSTM_MID_B in csrc/lstm.hip was set from
this sweep run with either kernel forced)."""
import sys, time
sys.path.insert(0, '.')
import torch
from em_pose_amd import _lib
from em_pose_amd.nn.layers import RNNLayer
for kv in sys.argv[1:]:          # NAME=INT kernel-variant switches, e.g. lstm_mid_x3=0
    k, _, v = kv.partition('=')
    _lib.check(_lib.lib().empose_set_option(k.encode(), int(v)))
dev = 'cuda:0'
layer = RNNLayer(60, 512, 2).eval().to(dev)
F = 64
for B in (9, 12, 16, 17, 32, 36, 64, 96, 128, 192, 256, 512):
    x = torch.randn(B, F, 60, device=dev)
    lens = torch.full((B,), F, device=dev)
    for _ in range(3): layer(x, lens)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(10): layer(x, lens)
    torch.cuda.synchronize(); ms = (time.time() - t0) / 10 * 1e3
    print('B=%3d F=%d: %.3f ms  %.2f us/step' % (B, F, ms, ms * 1e3 / (F + 1)))
