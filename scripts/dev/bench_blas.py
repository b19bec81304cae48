"""Dev: the vendor library (hipBLASLt/rocBLAS through torch.mm, fp32) on the update-net hidden-layer shape, as a yardstick."""
import torch
dev = torch.device('cuda:0')
torch.backends.cuda.matmul.allow_tf32 = False
for (M, N, K) in [(65536, 512, 512), (32768, 512, 512), (32768, 2048, 656), (65536, 512, 296)]:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev)
    for _ in range(3): C = A @ W.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record()
        for _ in range(10): C = torch.mm(A, W.t())
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    t = sorted(ts)[2]
    print('torch.mm fp32 M=%d N=%d K=%d: %.1f us  %.1f TFLOP/s' % (M, N, K, t * 1e3, 2.0 * M * N * K / t / 1e9))
