// Kernels of the training backward pass (BASELINE.json configs[4]; reference nn/models.py:634-688 runs it through
// torch.autograd): what the forward kernels do not already cover.
//
//   gemm_atb     C[N][K] = A[M][N]^T . B[M][K]  -- the weight gradient of a linear layer (dW = dY^T X) and of the LSTM
//                (dW_ih = dG^T X, dW_hh = dG^T H_prev), optionally with the column sums of A (the bias gradient).
//                Both operands are read as they lie in memory (row-major activations, the reduction index m is the row):
//                for v_mfma_f32_32x32x2_f32 the A operand wants A^T[n][m] per lane (n = lane & 31, m parity = lane >> 5)
//                and the B operand B[m][k] (k = lane & 31), i.e. two coalesced 128-byte rows per half-wave straight from
//                global memory into registers -- no LDS, no transposed copy.  The output has few tiles (512 x 512 = 16
//                tiles of 128 x 128) and a long reduction (M = frames), so M is split over blockIdx.y; partial tiles go
//                to a workspace and are summed in split order by a second kernel (deterministic).
//   transpose    W[N][K] -> W^T[K][N]: dX = dY . W runs on the forward GEMM kernels (A . W'^T with W' = W^T).
//   lstm_cell_bwd  one time step of back-propagation through time: gate pre-activation gradients from (dh, dc) and the
//                activations saved by the forward kernels (LstmUnitArgs::sv_*); the recurrent product dh_{t-1} = dG_t W_hh
//                is a GEMM launch per step.
//   lgd_losses   the loss terms of IterativeErrorFeedback.backward (reference models.py:634-688, loss.py:13-41) and their
//                cotangents in one pass.
#include "kernels.h"

#include <algorithm>

namespace empose {

typedef float f32x16t __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------------------------
// C = A^T B
// ---------------------------------------------------------------------------------------------------------------
namespace atb {
constexpr int NT = 256;            // 4 waves: 2 x 2 wave tiles of 64 x 64
constexpr int BN = 128, BK = 128;
}  // namespace atb

__global__ __launch_bounds__(atb::NT) void gemm_atb_kernel(AtbArgs a) {
  using namespace atb;
  const int tiles_k = (a.K + BK - 1) / BK;
  const int tn = blockIdx.x / tiles_k, tk = blockIdx.x - tn * tiles_k;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int n0 = tn * BN + (wave >> 1) * 64, k0 = tk * BK + (wave & 1) * 64;
  // rows of this split: chunks of 8 rows (four k-pairs of the 32x32x2 instruction per trip)
  const int chunk = (((a.M + a.S - 1) / a.S) + 7) & ~7;
  const int ms = blockIdx.y * chunk, me = min(a.M, ms + chunk);
  f32x16t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const bool nv[2] = {n0 + l31 < a.N, n0 + 32 + l31 < a.N};
  const bool kv[2] = {k0 + l31 < a.K, k0 + 32 + l31 < a.K};
  const float* pa = a.A + n0 + l31;
  const float* pb = a.B + k0 + l31;
  float bsum[2] = {0.f, 0.f};
  const bool do_bias = a.bias_partial != nullptr && tk == 0 && (wave & 1) == 0;
  for (int m = ms; m < me; m += 8) {
    float fa[4][2], fb[4][2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int mm = m + 2 * q + lh;
      const bool ok = mm < me;
      const size_t ra = (size_t)mm * a.lda, rb = (size_t)mm * a.ldb;
      fa[q][0] = ok && nv[0] ? pa[ra] : 0.f;
      fa[q][1] = ok && nv[1] ? pa[ra + 32] : 0.f;
      fb[q][0] = ok && kv[0] ? pb[rb] : 0.f;
      fb[q][1] = ok && kv[1] ? pb[rb + 32] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][i], fb[q][j], acc[i][j], 0, 0, 0);
      if (do_bias) { bsum[0] += fa[q][0]; bsum[1] += fa[q][1]; }
    }
  }
  // C/D layout of the 32x32 instruction: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  float* out = a.S > 1 ? a.partial + (size_t)blockIdx.y * a.N * a.K : a.C;
  const int ldo = a.S > 1 ? a.K : a.ldc;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = k0 + j * 32 + l31;
      if (k >= a.K) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (n < a.N) out[(size_t)n * ldo + k] = acc[i][j][r];
      }
    }
  if (do_bias) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float v = bsum[i] + __shfl_xor(bsum[i], 32, 64);   // the two row parities
      const int n = n0 + i * 32 + l31;
      if (lh == 0 && n < a.N) (a.S > 1 ? a.bias_partial + (size_t)blockIdx.y * a.N : a.bias)[n] = v;
    }
  }
}

// C[i] = sum_s partial[s][i] in split order; the same for the bias partials
__global__ void atb_reduce_kernel(AtbArgs a) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nk = (size_t)a.N * a.K;
  if (i < nk) {
    float v = 0.f;
    for (int s = 0; s < a.S; ++s) v += a.partial[(size_t)s * nk + i];
    const size_t n = i / a.K, k = i - n * a.K;
    a.C[n * a.ldc + k] = v;
  }
  if (a.bias_partial && i < (size_t)a.N) {
    float v = 0.f;
    for (int s = 0; s < a.S; ++s) v += a.bias_partial[(size_t)s * a.N + i];
    a.bias[i] = v;
  }
}

int atb_splits(int M, int N, int K) {
  const int tiles = ((N + atb::BN - 1) / atb::BN) * ((K + atb::BK - 1) / atb::BK);
  int s = (768 + tiles - 1) / tiles;
  s = std::min(s, std::max(1, M / 64));
  return std::max(1, std::min(s, 256));
}

size_t atb_workspace_floats(int M, int N, int K) {
  const int s = atb_splits(M, N, K);
  return s > 1 ? (size_t)s * ((size_t)N * K + N) : 0;
}

hipError_t launch_gemm_atb(AtbArgs a, float* workspace, hipStream_t stream) {
  a.S = atb_splits(a.M, a.N, a.K);
  a.partial = nullptr;
  a.bias_partial = a.bias;   // S == 1: the kernel writes the bias directly (pointer doubles as the flag)
  if (a.S > 1) {
    a.partial = workspace;
    a.bias_partial = a.bias ? workspace + (size_t)a.S * a.N * a.K : nullptr;
  }
  const int tiles = ((a.N + atb::BN - 1) / atb::BN) * ((a.K + atb::BK - 1) / atb::BK);
  hipLaunchKernelGGL(gemm_atb_kernel, dim3(tiles, a.S), dim3(atb::NT), 0, stream, a);
  if (a.S > 1) {
    const size_t n = (size_t)a.N * a.K;
    hipLaunchKernelGGL(atb_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a);
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const float* src, int ld_src, float* dst, int ld_dst, int rows,
                                                        int cols) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + ty + 8 * k, c = c0 + tx;
    tile[ty + 8 * k][tx] = (r < rows && c < cols) ? src[(size_t)r * ld_src + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + ty + 8 * k, r = r0 + tx;
    if (c < cols && r < rows) dst[(size_t)c * ld_dst + r] = tile[tx][ty + 8 * k];
  }
}

hipError_t launch_transpose(const float* src, int ld_src, float* dst, int ld_dst, int rows, int cols,
                            hipStream_t stream) {
  hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, stream, src, ld_src, dst,
                     ld_dst, rows, cols);
  return hipGetLastError();
}

// out[i] = a[i] + b[i]  (b_ih + b_hh)
__global__ void add2_kernel(const float* a, const float* b, float* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[i] + b[i];
}
hipError_t launch_add2(const float* a, const float* b, float* out, int n, hipStream_t stream) {
  hipLaunchKernelGGL(add2_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, a, b, out, n);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// One step of back-propagation through time of an LSTM layer (gate order i, f, g, o; PyTorch semantics for packed ragged
// rows: a step at or past a row's length does not exist -- its state passes through, its output is zero).
//   dh = dy[b][t] + dh_in[b]     (dh_in: from step t + 1, incl. the pass-through of rows that had ended there)
//   live:   do = dh tanh(c_t); dc = dc_in + dh o (1 - tanh(c_t)^2); di = dc g; dg = dc i; df = dc c_{t-1}
//           dG = (di i(1-i), df f(1-f), dg (1-g^2), do o(1-o));  dc_out = dc f;  carry = 0
//   ended:  dG = 0;  dc_out = dc_in;  carry = dh_in
// The caller then launches dh_in' = dG_t . W_hh + carry.
// ---------------------------------------------------------------------------------------------------------------
__global__ void lstm_cell_bwd_kernel(LstmCellBwdArgs a) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= a.B * a.H) return;
  const int b = idx / a.H, j = idx - b * a.H;
  const int H = a.H, F = a.F, t = a.t;
  const int len = a.seq_lengths ? a.seq_lengths[b] : F;
  const size_t row = (size_t)b * F + t;
  float* dG = a.dgates + row * 4 * H + j;
  const float dh_in = a.dh_in ? a.dh_in[idx] : 0.f;
  const float dc_in = a.dc[idx];
  if (t >= len) {
    dG[0] = 0.f; dG[H] = 0.f; dG[2 * H] = 0.f; dG[3 * H] = 0.f;
    a.dh_carry[idx] = dh_in;
    return;   // dc passes through unchanged
  }
  const float* g4 = a.gates + row * 4 * H + j;
  const float gi = g4[0], gf = g4[H], gg = g4[2 * H], go = g4[3 * H];
  const float c_t = a.c_all[row * H + j];
  const float c_prev = t > 0 ? a.c_all[(row - 1) * H + j] : (a.c0 ? a.c0[idx] : 0.f);
  const float dh = (a.dy ? a.dy[row * a.ld_dy + j] : 0.f) + dh_in;
  const float tc = tanhf(c_t);
  const float d_o = dh * tc;
  const float dc = dc_in + dh * go * (1.f - tc * tc);
  dG[0] = dc * gg * gi * (1.f - gi);
  dG[H] = dc * c_prev * gf * (1.f - gf);
  dG[2 * H] = dc * gi * (1.f - gg * gg);
  dG[3 * H] = d_o * go * (1.f - go);
  a.dc[idx] = dc * gf;
  a.dh_carry[idx] = 0.f;
}

hipError_t launch_lstm_cell_bwd(const LstmCellBwdArgs& a, hipStream_t stream) {
  const int n = a.B * a.H;
  hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, a);
  return hipGetLastError();
}

}  // namespace empose
