// One time step of an LSTM layer on the fp32 matrix cores (reference nn/layers.py:133-157, nn.LSTM gate order
// i,f,g,o).  The input projection x_t . W_ih^T + b_ih + b_hh for ALL steps is one plain GEMM (gemm_f32.hip); this
// kernel adds the recurrent term h_{t-1} . W_hh^T and applies the cell non-linearities in its epilogue.
//
// Tiling: a wave owns 32 batch rows x 32 hidden units and keeps FOUR 32x32 accumulators, one per gate, so that
// i/f/g/o of one (row, unit) land in the same lane and the cell update needs no cross-lane traffic.  A block is
// 2x2 waves = 64 rows x 64 units (x 4 gates = 256 W_hh rows staged per K tile).
// Ragged windows (pack_padded_sequence semantics): rows with t >= seq_length keep (h, c) and emit zeros.
#include "kernels.h"

namespace empose {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int LBK = 32;
constexpr int LLD = LBK + 4;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(256) void lstm_step_kernel(LstmStepArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[(64 + 256) * LLD];
  float* As = lds;             // [64 rows][LLD]
  float* Bs = lds + 64 * LLD;  // [4 gates][64 units][LLD]

  const int H = a.H;
  const int m0 = blockIdx.x * 64;
  const int j0 = blockIdx.y * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wrow = wave >> 1, wcol = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;

  f32x16 acc[4];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;

  const int nk = (H + LBK - 1) / LBK;
  for (int kt = 0; kt < nk; ++kt) {
    const int k0 = kt * LBK;
    // stage h_prev rows (64 x 32) and the 4 x 64 W_hh rows of this unit tile
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int slot = tid + i * 256;
      const int r = slot >> 3, c4 = (slot & 7) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m0 + r < a.B && k0 + c4 < H) v = *reinterpret_cast<const float4*>(a.h_prev + (size_t)(m0 + r) * H + k0 + c4);
      *reinterpret_cast<float4*>(As + r * LLD + c4) = v;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int slot = tid + i * 256;
      const int r = slot >> 3, c4 = (slot & 7) * 4;  // r in [0,256): gate = r >> 6, unit = r & 63
      const int unit = j0 + (r & 63);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (unit < H && k0 + c4 < H)
        v = *reinterpret_cast<const float4*>(a.w_hh + (size_t)((r >> 6) * H + unit) * H + k0 + c4);
      *reinterpret_cast<float4*>(Bs + r * LLD + c4) = v;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < LBK / 8; ++kk) {
      const float4 av = *reinterpret_cast<const float4*>(As + (wrow * 32 + l31) * LLD + kk * 8 + lh * 4);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 bv = *reinterpret_cast<const float4*>(Bs + (g * 64 + wcol * 32 + l31) * LLD + kk * 8 + lh * 4);
        acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[g], 0, 0, 0);
        acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[g], 0, 0, 0);
        acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[g], 0, 0, 0);
        acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[g], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  const int unit = j0 + wcol * 32 + l31;
  if (unit >= H) return;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = m0 + wrow * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    if (row >= a.B) continue;
    const size_t hc = (size_t)row * H + unit;
    const size_t yo = ((size_t)row * a.F + a.t) * H + unit;
    const bool live = a.seq_lengths ? (a.t < a.seq_lengths[row]) : true;
    if (!live) {
      a.h_next[hc] = a.h_prev[hc];
      a.y[yo] = 0.f;
      continue;
    }
    const float* gin = a.gin + ((size_t)row * a.F + a.t) * 4 * H + unit;
    const float gi = acc[0][r] + gin[0];
    const float gf = acc[1][r] + gin[H];
    const float gg = acc[2][r] + gin[2 * H];
    const float go = acc[3][r] + gin[3 * H];
    const float c_new = sigmoidf_(gf) * a.c[hc] + sigmoidf_(gi) * tanhf(gg);
    const float h_new = sigmoidf_(go) * tanhf(c_new);
    a.c[hc] = c_new;
    a.h_next[hc] = h_new;
    a.y[yo] = h_new;
  }
}

hipError_t launch_lstm_step(const LstmStepArgs& a, hipStream_t stream) {
  dim3 grid((a.B + 63) / 64, (a.H + 63) / 64);
  hipLaunchKernelGGL(lstm_step_kernel, grid, dim3(256), 0, stream, a);
  return hipGetLastError();
}

}  // namespace empose
