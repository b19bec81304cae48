"""Dev: empose_linear_f32 on the small problems of the training step at the reference's batch (12 windows = 384 rows)."""
import sys, time
sys.path.insert(0, '.')
import torch
from em_pose_amd import _lib
lib = _lib.lib(); dev = 'cuda:0'
for M, N, K in ((384, 512, 512), (384, 512, 296), (384, 66, 512), (384, 296, 512), (12, 512, 2048), (12, 2048, 512),
                (384, 2048, 144), (384, 2048, 512), (256, 512, 2048), (64, 512, 2048), (8192, 512, 512)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); y = torch.empty(M, N, device=dev)
    def run():
        _lib.check(lib.empose_linear_f32(_lib.dptr(x), K, _lib.dptr(w), K, _lib.dptr(y), N, M, N, K, None, None, 0, 0.0, None))
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 200 * 1e3
    print('M=%5d N=%5d K=%5d: %6.1f us per call back to back, %5.1f TFLOP/s' % (M, N, K, us, 2.0 * M * N * K / us * 1e-6))
