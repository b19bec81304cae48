#include "../../em_pose_amd/csrc/mlp_fused.hip"
#include <cstdio>
using namespace empose;
__global__ void fill_kernel(float* p, size_t n, unsigned seed, float scale) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i >= n) return;
  unsigned x = (unsigned)i * 2654435761u ^ seed; x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15;
  p[i] = ((x & 0xffffff) / 8388608.f - 1.f) * scale;
}
static float* dev_rand(size_t n, unsigned seed, float scale) { float* p; (void)hipMalloc(&p, n * 4); fill_kernel<<<(n + 255) / 256, 256>>>(p, n, seed, scale); return p; }
int main() {
  const int T = 32768;
  int shapes[2][2] = {{200, 320}, {320, 200}};
  for (auto& sh : shapes) {
    const int K = sh[0], N = sh[1];
    FusedMlpArgs a; a.count = 1; a.M = T;
    FusedNet& fn = a.net[0];
    fn.x = dev_rand((size_t)T * K, 1, 1.f); fn.ldx = K; fn.out = dev_rand((size_t)T * N, 0, 0.f); fn.ld_out = N; fn.n_layers = 1;
    FusedLayer& L = fn.layer[0];
    L.K = K; L.N = N; L.W = dev_rand((size_t)((((K + 7) / 8) + 3) & ~3) * ((N + 31) / 32) * 256, 3, 0.05f);
    L.scale = nullptr; L.shift = nullptr; L.slope = 0.f; L.act = 0;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) (void)launch_mlp_fused(a, 0);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) (void)launch_mlp_fused(a, 0);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    printf("K=%d N=%d: %.1f us  %.1f TFLOP/s (%s)\n", K, N, ms * 1e3, 2.0 * T * N * K / ms * 1e-9, hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
