// Dev lab: candidate fp32 GEMM kernels against the production one (em_pose_amd/csrc/gemm_f32.hip), no torch.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iem_pose_amd/csrc scripts/dev/gemm_lab.hip -o /tmp/gemm_lab && /tmp/gemm_lab
#define EMPOSE_GEMM_TRACE 1
#include "../../em_pose_amd/csrc/gemm_f32.hip"
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
  float* p; hipMalloc(&p, n * 4);
  fill_kernel<<<(n + 255) / 256, 256>>>(p, n, seed, scale);
  return p;
}

template <typename F>
static float time_ms(F f, int reps = 20) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) f();
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms / reps < best ? ms / reps : best;
  }
  return best;
}

int main(int argc, char** argv) {
  struct Shape { int M, N, K, nprob; };
  std::vector<Shape> shapes = {{32768, 512, 512, 2}, {32768, 512, 296, 2}, {32768, 2048, 512, 1}, {32768, 320, 200, 1},
                               {32768, 200, 320, 1}, {8192, 512, 512, 2}, {32768, 66, 512, 2}};
  // usage: gemm_lab [shape_index [variant_index]]  (no arguments: everything)
  const int only_shape = argc > 1 ? atoi(argv[1]) : -1, only_variant = argc > 2 ? atoi(argv[2]) : -1;
  for (size_t si = 0; si < shapes.size(); ++si) {
    if (only_shape >= 0 && (int)si != only_shape) continue;
    const Shape& s = shapes[si];
    GemmBatch b;
    b.count = s.nprob; b.role = 1;
    std::vector<float*> Cs, Cr;
    for (int i = 0; i < s.nprob; ++i) {
      GemmProb& p = b.p[i];
      p.A = dev_rand((size_t)s.M * s.K, 11 + i, 1.f); p.lda = s.K;
      p.W = dev_rand((size_t)s.N * s.K, 23 + i, 0.05f); p.ldw = s.K;
      p.C = dev_rand((size_t)s.M * s.N, 0, 0.f); p.ldc = s.N;
      p.M = s.M; p.N = s.N; p.K = s.K;
      p.scale = dev_rand(s.N, 5, 1.f); p.shift = dev_rand(s.N, 6, 1.f);
      p.resid = nullptr; p.ldr = 0; p.act = 1; p.slope = 0.25f;
      Cs.push_back(p.C);
      float* ref; hipMalloc(&ref, (size_t)s.M * s.N * 4); Cr.push_back(ref);
    }
    const double flops = 2.0 * s.M * s.N * s.K * s.nprob;
    // reference result: the small-tile configuration (independent code path of the same template)
    GemmBatch br = b;
    for (int i = 0; i < s.nprob; ++i) br.p[i].C = Cr[i];
    launch_cfg<Cfg<2, 2, 2, 2, 32, false>, 0>(br, 0);
    hipDeviceSynchronize();
    using CfgWdb = Cfg<2, 2, 4, 4, 32, true>;    // 256 x 256, 4 waves, wave tile 128 x 128, double-buffered LDS
    using CfgWsb = Cfg<2, 2, 4, 4, 32, false>;
    using CfgW16 = Cfg<2, 2, 4, 4, 16, true>;
    const char* names[] = {"CfgX", "dispatch", "W256db", "wide"};
    for (int variant = 0; variant < 4; ++variant) {
      if (only_variant >= 0 && variant != only_variant) continue;
      auto run = [&]() {
        switch (variant) {
          case 0: launch_cfg<CfgX, 1>(b, 0); break;
          case 1: launch_gemm(b, 0); break;
          case 2: launch_cfg<CfgWdb, 1>(b, 0); break;
          case 3: launch_wide<1>(b, 0); break;
        }
      };
      for (int i = 0; i < s.nprob; ++i) hipMemset(Cs[i], 0, (size_t)s.M * s.N * 4);
      float ms = time_ms(run);
      hipError_t e = hipDeviceSynchronize();
      // compare
      double maxd = 0;
      for (int i = 0; i < s.nprob; ++i) {
        std::vector<float> h1((size_t)s.M * s.N), h2((size_t)s.M * s.N);
        hipMemcpy(h1.data(), Cs[i], h1.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(h2.data(), Cr[i], h2.size() * 4, hipMemcpyDeviceToHost);
        for (size_t k = 0; k < h1.size(); ++k) { double d = fabs((double)h1[k] - h2[k]); if (!(d <= maxd)) maxd = d; }
      }
      if (variant == 3 && si == 0) {
        long long tr[2][64];
        hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_gemm_trace), sizeof(tr));
        const int nkk = (s.K + 31) / 32;
        for (int b2 = 0; b2 < 2; ++b2) {
          printf("  trace block (0,%d): prologue %lld |", b2, tr[b2][1] - tr[b2][0]);
          for (int k = 0; k < nkk; ++k) printf(" %lld", tr[b2][2 + k] - tr[b2][1 + k]);
          printf(" | epilogue %lld | total %lld | start at wall %lld (100 MHz ticks), shader clock %.0f MHz\n",
                 tr[b2][2 + nkk] - tr[b2][1 + nkk], tr[b2][2 + nkk] - tr[b2][0], tr[b2][62] - tr[0][62],
                 100.0 * (tr[b2][2 + nkk] - tr[b2][0]) / (double)(tr[b2][63] - tr[b2][62]));
        }
      }
      printf("M=%d N=%d K=%d x%d  %-10s %8.1f us  %6.1f TFLOP/s  maxdiff %.2e  %s\n", s.M, s.N, s.K, s.nprob,
             names[variant], ms * 1e3, flops / ms * 1e-9, maxd, e == hipSuccess ? "" : hipGetErrorString(e));
    }
    for (int i = 0; i < s.nprob; ++i) { hipFree((void*)b.p[i].A); hipFree((void*)b.p[i].W); hipFree(Cs[i]); hipFree(Cr[i]);
                                        hipFree((void*)b.p[i].scale); hipFree((void*)b.p[i].shift); }
  }
  return 0;
}
