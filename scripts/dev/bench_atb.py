"""Dev: the weight-gradient GEMM C = A^T B at the training shapes, for several split targets."""
import sys; sys.path.insert(0, '.')
import torch
from em_pose_amd import _lib
lib = _lib.lib(); dev = 'cuda:0'
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for M, N, K in ((32768, 512, 512), (32768, 512, 296), (16384, 512, 512), (8192, 512, 512), (8192, 512, 296), (4096, 512, 512), (8192, 2048, 512), (2048, 512, 512), (384, 512, 512), (8192, 66, 512)):
    A, B = torch.randn(M, N, device=dev), torch.randn(M, K, device=dev)
    Cm, bias = torch.empty(N, K, device=dev), torch.empty(N, device=dev)
    res = []
    for target, rows in ((256, 32), (0, 0), (256, 16), (512, 32), (512, 16), (1024, 16)):   # (0, 0): chosen by size
        _lib.check(lib.empose_set_option(b'atb_target', target))
        _lib.check(lib.empose_set_option(b'atb_chunk', rows))
        nb = lib.empose_gemm_atb_workspace_bytes(M, N, K); ws = torch.empty(max(nb, 4), dtype=torch.uint8, device=dev)
        us = t(lambda: lib.empose_gemm_atb_f32(M, N, K, A.data_ptr(), N, B.data_ptr(), K, Cm.data_ptr(), K, bias.data_ptr(), ws.data_ptr(), ws.numel(), None))
        res.append('%dx%d:%.0fus(%.0fTF)' % (target, rows, us, 2.0 * M * N * K / us / 1e6))
    print(M, N, K, ' '.join(res))
