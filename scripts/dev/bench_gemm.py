"""Dev: time empose_linear_f32 on the update-net hidden-layer shape (M=65536 rows = both nets, N=K=512)."""
import os, sys
sys.path.insert(0, '.')
import torch
from em_pose_amd import _lib
lib = _lib.lib()
dev = torch.device('cuda:0')
shapes = [(65536, 512, 512), (65536, 512, 296), (32768, 2048, 512), (32768, 320, 200), (32768, 200, 320), (65536, 66, 512)]
for (M, N, K) in shapes:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / K ** 0.5
    C = torch.empty(M, N, device=dev); sc = torch.rand(N, device=dev) + 0.5; sh = torch.randn(N, device=dev)
    def run():
        _lib.check(lib.empose_linear_f32(_lib.dptr(A), K, _lib.dptr(W), K, _lib.dptr(C), N, M, N, K, _lib.dptr(sc), _lib.dptr(sh), 1, 0.25, None))
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    t = sorted(ts)[len(ts) // 2]
    ref = torch.nn.functional.prelu((A[:256] @ W.t()) * sc + sh, torch.tensor([0.25], device=dev))
    err = (C[:256] - ref).abs().max().item()
    print('GEMM_VARIANT=%s M=%d N=%d K=%d: %.1f us  %.1f TFLOP/s  err %.1e' % (os.environ.get('EMPOSE_GEMM_VARIANT', '-'), M, N, K, t * 1e3, 2.0 * M * N * K / t / 1e9, err))
