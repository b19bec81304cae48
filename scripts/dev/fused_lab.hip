// Dev lab: the fused MLP kernel on the bench shape (two nets, 296-512x4-66/10, T rows), timing + optional phase stamps.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iem_pose_amd/csrc scripts/dev/fused_lab.hip -o /tmp/fused_lab && /tmp/fused_lab
#define EMPOSE_FUSED_TRACE 1
#include "../../em_pose_amd/csrc/mlp_fused.hip"
#include "lab_stubs.h"

#include <cstdio>
#include <vector>
using namespace empose;

__global__ void fill_kernel(float* p, size_t n, unsigned seed, float scale) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned x = (unsigned)i * 2654435761u ^ seed;
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  p[i] = ((x & 0xffffff) / 8388608.f - 1.f) * scale;
}
static float* dev_rand(size_t n, unsigned seed, float scale) {
  float* p; (void)hipMalloc(&p, n * 4);
  fill_kernel<<<(n + 255) / 256, 256>>>(p, n, seed, scale);
  return p;
}

int main(int argc, char** argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 32768, LDX = 296, Hd = 512;
  FusedMlpArgs a;
  a.count = 2; a.M = T;
  float* x = dev_rand((size_t)T * LDX, 1, 1.f);
  double flops = 0;
  for (int n = 0; n < 2; ++n) {
    FusedNet& fn = a.net[n];
    const int nout = n == 0 ? 66 : 10;
    fn.x = x; fn.ldx = LDX; fn.out = dev_rand((size_t)T * nout, 0, 0.f); fn.ld_out = nout;
    fn.n_layers = 6;
    for (int l = 0; l < 6; ++l) {
      FusedLayer& L = fn.layer[l];
      L.K = l == 0 ? LDX : Hd; L.N = l == 5 ? nout : Hd;
      L.W = dev_rand((size_t)((((L.K + 7) / 8) + 3) & ~3) * ((L.N + 31) / 32) * 256, 100 + 10 * n + l, 0.06f);   // fragment order
      L.scale = dev_rand(L.N, 7, 1.f); L.shift = dev_rand(L.N, 8, 0.1f);
      L.slope = 0.25f; L.act = l < 5;
      flops += 2.0 * T * L.N * L.K;
    }
  }
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) (void)launch_mlp_fused(a, 0);
  (void)hipDeviceSynchronize();
  float best = 1e9f;
  for (int r = 0; r < 3; ++r) {
    (void)hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) (void)launch_mlp_fused(a, 0);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    best = ms / 10 < best ? ms / 10 : best;
  }
  printf("T=%d: %.1f us/launch  %.1f TFLOP/s  (%s)\n", T, best * 1e3, flops / best * 1e-9, hipGetErrorString(hipGetLastError()));
  long long tr[64];
  (void)hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_fused_trace), sizeof(tr));
  printf("  block (0,0): ");
  for (int l = 0; l < 6; ++l)
    printf("L%d: prologue %lld, k-loop %lld (%.0f/tile), epilogue %lld | ", l, tr[1 + 4 * l] - tr[4 * l], tr[2 + 4 * l] - tr[1 + 4 * l],
           (double)(tr[2 + 4 * l] - tr[1 + 4 * l]) / ((l == 0 ? LDX + 31 : Hd) / 32), tr[3 + 4 * l] - tr[2 + 4 * l]);
  printf("total %lld\n", tr[3 + 4 * 5] - tr[0]);
  return 0;
}
