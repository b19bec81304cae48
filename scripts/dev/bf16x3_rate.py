"""Dev (GPU): how fast are the six bf16 products of a bf16x3-split layer next to the one fp32 product, with the vendor
library on both sides (hipBLASLt through torch.mm)?  An upper-level estimate for DESIGN.md 9.8, not a kernel."""
import time, torch
dev = 'cuda:0'
M, N, K = 32768, 512, 512
a32, w32 = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
a16 = [torch.randn(M, K, device=dev).bfloat16() for _ in range(3)]
w16 = [torch.randn(N, K, device=dev).bfloat16() for _ in range(3)]
def bench(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
t32 = bench(lambda: torch.mm(a32, w32.t()))
t16 = bench(lambda: torch.mm(a16[0], w16[0].t()))
pairs = [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)]
t6 = bench(lambda: [torch.mm(a16[i], w16[j].t()) for i, j in pairs])
flops = 2.0 * M * N * K
print('fp32 GEMM %d x %d x %d: %.1f us (%.0f TFLOP/s)' % (M, N, K, t32, flops / t32 * 1e-6))
print('one bf16 GEMM: %.1f us (%.0f TFLOP/s); six of them: %.1f us -> %.2fx the fp32 GEMM' % (t16, flops / t16 * 1e-6, t6, t32 / t6))
