// Dev lab: where the time of the weight-gradient GEMM C = A^T B (em_pose_amd/csrc/train.hip, gemm_atb_lds_kernel) goes.
// A stripped copy of its main loop (no bounds, no segments, no bias) with parts switched off at compile time:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/dev/atb_lab.hip -o scripts/dev/bin/atb_lab && scripts/dev/bin/atb_lab
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f32x16t __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));

// V bit 0: no global loads inside the loop   bit 1: no LDS writes   bit 2: no LDS reads (constant operands)
// V bit 3: global_load_lds (16 bytes per lane straight into LDS, no staging registers, no ds_write)
// TM x TK: wave tile in 32-blocks (2 x 2 = the production kernel: 64 x 64 per wave, 128 x 128 per workgroup)
template <int V, int MC>
__global__ __launch_bounds__(256) void atb_kernel(const float* A, const float* B, float* partial, int M, int N, int K,
                                                  int S) {
  constexpr int BN = 128, BK = 128;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tiles_k = K / BK;
  const int tn = blockIdx.x / tiles_k, tk = blockIdx.x - tn * tiles_k;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int nw = (wave >> 1) * 64, kw = (wave & 1) * 64;
  const int n_base = tn * BN, k_base = tk * BK;
  const int chunk = M / S;
  const int ms = blockIdx.y * chunk, me = ms + chunk;
  f32x16t acc[2][2];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int r8 = tid >> 5, c4 = (tid & 31) * 4;
  constexpr int P = MC / 8;
  f4 ga[P], gb[P];
  auto gload = [&](int m0) {
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int mm = m0 + r8 + 8 * p;
      ga[p] = *reinterpret_cast<const f4*>(A + (size_t)mm * N + n_base + c4);
      gb[p] = *reinterpret_cast<const f4*>(B + (size_t)mm * K + k_base + c4);
    }
  };
  auto lwrite = [&](float* st) {
#pragma unroll
    for (int p = 0; p < P; ++p) {
      *reinterpret_cast<f4*>(st + (r8 + 8 * p) * BN + c4) = ga[p];
      *reinterpret_cast<f4*>(st + MC * BN + (r8 + 8 * p) * BK + c4) = gb[p];
    }
  };
  constexpr int STAGE = MC * (BN + BK);
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* glb_ptr_t;
  auto gdma = [&](int m0, float* st) {   // a wave instruction fills 1 KB of LDS: its two rows of one operand
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int mm = m0 + r8 + 8 * p;
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(A + (size_t)mm * N + n_base + c4),
                                       (lds_ptr_t)(st + (2 * wave + 8 * p) * BN), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(B + (size_t)mm * K + k_base + c4),
                                       (lds_ptr_t)(st + MC * BN + (2 * wave + 8 * p) * BK), 16, 0, 0);
    }
  };
  if (V & 8) {
    gdma(ms, lds);
  } else {
    gload(ms);
    lwrite(lds);
  }
  __syncthreads();
  int buf = 0;
  for (int m = ms; m < me; m += MC) {
    const bool more = m + MC < me;
    if (more && (V & 8)) gdma(m + MC, lds + (buf ^ 1) * STAGE);
    if (more && !(V & 1) && !(V & 8)) gload(m + MC);
    const float* sA = lds + buf * STAGE + nw + l31;
    const float* sB = lds + buf * STAGE + MC * BN + kw + l31;
    float a0 = sA[lh * BN], a1 = sA[lh * BN + 32];
    float b0 = sB[lh * BK], b1 = sB[lh * BK + 32];
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
    for (int q = 0; q < MC / 2; ++q) {
      const int row = (q + 1 < MC / 2 ? 2 * (q + 1) : 0) + lh;
      float na0 = a0, na1 = a1, nb0 = b0, nb1 = b1;
      if (!(V & 4)) {
        na0 = sA[row * BN]; na1 = sA[row * BN + 32];
        nb0 = sB[row * BK]; nb1 = sB[row * BK + 32];
      }
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      if (!(V & 4)) {
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      }
      a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
    }
    if (more && !(V & 2) && !(V & 8)) lwrite(lds + (buf ^ 1) * STAGE);
    __syncthreads();
    buf ^= 1;
  }
  float* out = partial + (size_t)blockIdx.y * N * K;
  const int n0 = n_base + nw, k0 = k_base + kw;
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        out[(size_t)n * K + k0 + j * 32 + l31] = acc[i][j][r];
      }
}

template <int V, int MC>
static void run(const char* name, const float* A, const float* B, float* partial, int M, int N, int K, int S) {
  const size_t lds = (size_t)2 * MC * 256 * sizeof(float);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&atb_kernel<V, MC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const dim3 grid((N / 128) * (K / 128), S);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((atb_kernel<V, MC>), grid, dim3(256), lds, 0, A, B, partial, M, N, K, S);
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((atb_kernel<V, MC>), grid, dim3(256), lds, 0, A, B, partial, M, N, K, S);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms / 20 < best ? ms / 20 : best;
  }
  printf("  %-44s S=%3d MC=%2d  %7.1f us  %6.1f TFLOP/s  (%s)\n", name, S, MC, best * 1e3, 2.0 * M * N * K / best / 1e9,
         hipGetErrorString(hipGetLastError()));
}

__global__ void fill_kernel(float* p, size_t n, unsigned seed) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned x = (unsigned)i * 2654435761u ^ seed;
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  p[i] = (x & 0xffffff) / 8388608.f - 1.f;   // random operands: the matrix cores draw more power than on zeros
}

int main() {
  const int M = 32768, N = 512, K = 512;
  float *A, *B, *partial;
  hipMalloc(&A, (size_t)M * N * 4); hipMalloc(&B, (size_t)M * K * 4); hipMalloc(&partial, (size_t)64 * N * K * 4);
  fill_kernel<<<(unsigned)(((size_t)M * N + 255) / 256), 256>>>(A, (size_t)M * N, 1u);
  fill_kernel<<<(unsigned)(((size_t)M * K + 255) / 256), 256>>>(B, (size_t)M * K, 2u);
  for (int S : {16, 32, 64}) {
    run<0, 32>("as in production", A, B, partial, M, N, K, S);
    run<1, 32>("no global loads in the loop", A, B, partial, M, N, K, S);
    run<3, 32>("no global loads, no LDS writes", A, B, partial, M, N, K, S);
    run<4, 32>("no LDS reads", A, B, partial, M, N, K, S);
    run<7, 32>("MFMAs + barriers only", A, B, partial, M, N, K, S);
    run<0, 64>("64-row chunks", A, B, partial, M, N, K, S);
    run<0, 16>("16-row chunks", A, B, partial, M, N, K, S);
    run<8, 32>("global_load_lds", A, B, partial, M, N, K, S);
    run<8, 16>("global_load_lds, 16-row chunks", A, B, partial, M, N, K, S);
    run<8, 64>("global_load_lds, 64-row chunks", A, B, partial, M, N, K, S);
  }
  // same result with and without the direct loads?
  float* p2; hipMalloc(&p2, (size_t)16 * N * K * 4);
  {
    const size_t lds = (size_t)2 * 32 * 256 * sizeof(float);
    const dim3 grid((N / 128) * (K / 128), 16);
    hipLaunchKernelGGL((atb_kernel<0, 32>), grid, dim3(256), lds, 0, A, B, partial, M, N, K, 16);
    hipLaunchKernelGGL((atb_kernel<8, 32>), grid, dim3(256), lds, 0, A, B, p2, M, N, K, 16);
    hipDeviceSynchronize();
    const size_t n = (size_t)16 * N * K;
    float* h0 = new float[n]; float* h1 = new float[n];
    hipMemcpy(h0, partial, n * 4, hipMemcpyDeviceToHost); hipMemcpy(h1, p2, n * 4, hipMemcpyDeviceToHost);
    double worst = 0; for (size_t i = 0; i < n; ++i) { double d = h0[i] - h1[i]; if (d < 0) d = -d; if (d > worst) worst = d; }
    printf("max |staged - direct| = %g  (sample %g)\n", worst, h0[12345]);
  }
  return 0;
}
