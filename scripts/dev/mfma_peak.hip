// Practical fp32 matrix-core ceiling of this GPU: a register-only MFMA loop (no LDS, no global memory), timed with HIP
// events, plus the shader clock it sustains (clock64 ticks per 100 MHz wall_clock64 tick).
//   hipcc --offload-arch=gfx950 -O3 scripts/dev/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(256) void spin(float* out, long long* clk, int iters) {
  const float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
  long long c0 = clock64(), w0 = wall_clock64();
  float s = 0.f;
  if (KIND == 0) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  } else {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j];
  }
  long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND>
void run(const char* name, int waves_per_simd, int iters) {
  const int blocks = 256 * waves_per_simd;
  float* out; long long* clk;
  hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, blocks * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  spin<KIND><<<blocks, 256>>>(out, clk, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  spin<KIND><<<blocks, 256>>>(out, clk, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_wave = (double)iters * (KIND == 0 ? 16 : 32);
  const double flops = mfma_per_wave * (KIND == 0 ? 32.0 * 32 * 2 * 2 : 16.0 * 16 * 4 * 2) * blocks * 4;
  long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  printf("%-14s waves/SIMD %d: %.3f ms  %.1f TFLOP/s  shader clock %.0f MHz (wall_clock64 at 100 MHz)\n", name,
         waves_per_simd, ms, flops / ms * 1e-9, 100.0 * h[0] / h[1]);
  hipFree(out); hipFree(clk);
}

int main() {
  for (int w = 1; w <= 2; ++w) {
    run<0>("32x32x2 f32", w, 20000);
    run<1>("16x16x4 f32", w, 20000);
  }
  run<0>("32x32x2 long", 2, 200000);
  return 0;
}
