import sys; sys.path.insert(0, '.')
import torch
from em_pose_amd import _lib
lib = _lib.lib(); dev = 'cuda:0'
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
M, N, K = 32768, 512, 512
A, B = torch.randn(M, N, device=dev), torch.randn(M, K, device=dev)
Cm, bias = torch.empty(N, K, device=dev), torch.empty(N, device=dev)
nb = lib.empose_gemm_atb_workspace_bytes(M, N, K); ws = torch.empty(max(nb, 4), dtype=torch.uint8, device=dev)
for rep in range(3):
    for name, bp in (('bias', bias.data_ptr()), ('no bias', None)):
        us = t(lambda: lib.empose_gemm_atb_f32(M, N, K, A.data_ptr(), N, B.data_ptr(), K, Cm.data_ptr(), K, bp, ws.data_ptr(), ws.numel(), None))
        print(name, '%.1f us' % us)
