// Dev lab: one wavefront launch of the LSTM kernel (both layers active) at the bench shape, with phase stamps.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iem_pose_amd/csrc scripts/dev/lstm_lab.hip -o /tmp/lstm_lab && /tmp/lstm_lab
#define EMPOSE_LSTM_TRACE 1
#include "../../em_pose_amd/csrc/lstm.hip"
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
  const int B = argc > 1 ? atoi(argv[1]) : 1024, F = 32, H = 512, IN = 144, LDX = 296;
  LstmWaveArgs a;
  a.n_units = 2; a.seq_lengths = nullptr; a.B = B; a.F = F; a.H = H;
  float* x = dev_rand((size_t)B * F * LDX, 1, 1.f);
  float* y = dev_rand((size_t)B * F * H, 0, 0.f);
  for (int l = 0; l < 2; ++l) {
    LstmUnitArgs& u = a.unit[l];
    u.in_k = l == 0 ? IN : H;
    u.w_ih = dev_rand((size_t)4 * H * u.in_k, 10 + l, 0.04f);
    u.w_hh = dev_rand((size_t)4 * H * H, 20 + l, 0.04f);
    u.bias = dev_rand(4 * H, 30 + l, 0.1f);
    u.in_seq = l == 0 ? x : nullptr; u.in_ld = LDX; u.in_from = l == 0 ? -1 : 0;
    u.t_offset = l; u.reverse = 0;
    u.h[0] = dev_rand((size_t)B * H, 40 + l, 0.5f); u.h[1] = dev_rand((size_t)B * H, 50 + l, 0.5f);
    u.c = dev_rand((size_t)B * H, 60 + l, 0.5f);
    u.y = l == 1 ? y : nullptr; u.y_ld = H; u.y_col = 0;
  }
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int variant = 0; variant < 2; ++variant) {
    a.s = 5;
    for (int i = 0; i < 5; ++i) { a.s = 5 + i; (void)launch_lstm_wave(a, 0); }
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
      (void)hipEventRecord(e0);
      for (int i = 0; i < 20; ++i) { a.s = 3 + i; (void)launch_lstm_wave(a, 0); }
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      best = ms / 20 < best ? ms / 20 : best;
    }
    const double flops = 2.0 * B * 4 * H * (IN + H + 2 * H);
    printf("B=%d: %.1f us/launch  %.1f TFLOP/s  (%s)\n", B, best * 1e3, flops / best * 1e-9, hipGetErrorString(hipGetLastError()));
    long long tr[128];
    (void)hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_lstm_trace), sizeof(tr));
    printf("  block (0,0,0): setup %lld  prologue %lld | tiles:", tr[1] - tr[0], tr[2] - tr[1]);
    const int n0 = (IN + 63) / 64 + H / 64, n1 = 2 * H / 64;
    int i = 3;
    for (int k = 0; k < n0; ++k, ++i) printf(" %lld", tr[i] - tr[i - 1]);
    printf(" | finish0 %lld | tiles:", tr[i] - tr[i - 1]); ++i;
    for (int k = 0; k < n1; ++k, ++i) printf(" %lld", tr[i] - tr[i - 1]);
    printf(" | finish1 %lld | total %lld\n", tr[i] - tr[i - 1], tr[i] - tr[0]);
    break;
  }
  return 0;
}
