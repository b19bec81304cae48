"""Dev: empose_linear_f32 on small problems (us per call, back to back): EMPOSE_GEMM_SPLITK=0 disables the split-K kernel."""
import sys, time
sys.path.insert(0, '.')
import torch
from em_pose_amd import _lib
lib = _lib.lib(); dev = 'cuda:0'
for (M, N, K) in [(32, 512, 512), (64, 512, 512), (256, 512, 512), (512, 512, 512), (1024, 512, 512), (2048, 512, 512),
                  (256, 512, 296), (256, 66, 512), (256, 10, 512), (256, 320, 200), (256, 200, 320), (32, 66, 512)]:
    x, w, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    def run():
        _lib.check(lib.empose_linear_f32_ex(_lib.dptr(x), K, _lib.dptr(w), K, _lib.dptr(out), N, M, N, K, None, _lib.dptr(b),
                                            None, 0, 1, 0.25, None))
    for _ in range(5): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): run()
    torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 200 * 1e6
    print('M=%5d N=%4d K=%4d: %6.1f us  %s' % (M, N, K, us, lib.empose_profile_gemm_kernel_name(M, N, K, 1, 0).decode()))
