// Dev lab: the two SMPL blend-shape GEMM shapes on the row-block kernel configurations vs the generic tile.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iem_pose_amd/csrc scripts/dev/blend_lab.hip -o /tmp/blend_lab && /tmp/blend_lab
#include "../../em_pose_amd/csrc/mlp_fused.hip"
#include "../../em_pose_amd/csrc/gemm_f32.hip"
#include "lab_stubs.h"
#include <cstdio>
using namespace empose;
__global__ void fill_kernel(float* p, size_t n, unsigned seed, float scale) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i >= n) return;
  unsigned x = (unsigned)i * 2654435761u ^ seed; x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15;
  p[i] = ((x & 0xffffff) / 8388608.f - 1.f) * scale;
}
static float* dev_rand(size_t n, unsigned seed, float scale) { float* p; (void)hipMalloc(&p, n * 4); fill_kernel<<<(n + 255) / 256, 256>>>(p, n, seed, scale); return p; }
template <typename F> static float time_us(F f) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) f();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) f();
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / 20 * 1e3f;
}
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
    GemmBatch b; b.count = 1;
    GemmProb& p = b.p[0];
    p.A = fn.x; p.lda = K; p.W = dev_rand((size_t)N * K, 4, 0.05f); p.ldw = K; p.C = fn.out; p.ldc = N; p.M = T; p.N = N; p.K = K;
    p.scale = nullptr; p.shift = nullptr; p.resid = nullptr; p.ldr = 0; p.act = 0; p.slope = 0.f;
    const double gf = 2.0 * T * N * K * 1e-3;
    float t;
    t = time_us([&]() { (void)launch_gemm(b, 0); });                                   printf("K=%d N=%d generic tile      %.1f us %.1f TF\n", K, N, t, gf / t);
    if (K <= 288) { t = time_us([&]() { (void)launch_gemm_rows_cfg<128, 2, 5, 2>(a, 0); }); printf("K=%d N=%d rows<128,2,5,2>  %.1f us %.1f TF\n", K, N, t, gf / t); }
    t = time_us([&]() { (void)launch_gemm_rows_cfg<64, 1, 4, 2>(a, 0); });             printf("K=%d N=%d rows<64,1,4,2>   %.1f us %.1f TF\n", K, N, t, gf / t);
    t = time_us([&]() { (void)launch_gemm_rows_cfg<64, 2, 2, 4>(a, 0); });             printf("K=%d N=%d rows<64,2,2,4>   %.1f us %.1f TF\n", K, N, t, gf / t);
    t = time_us([&]() { (void)launch_gemm_rows_cfg<64, 2, 3, 4>(a, 0); });             printf("K=%d N=%d rows<64,2,3,4>   %.1f us %.1f TF\n", K, N, t, gf / t);
    printf("  (%s)\n", hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
