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
#include "lstm_cell_bwd.h"

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
  const bool acc_c = a.S == 1 && a.accumulate;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = k0 + j * 32 + l31;
      if (k >= a.K) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (n < a.N) out[(size_t)n * ldo + k] = acc[i][j][r] + (acc_c ? out[(size_t)n * ldo + k] : 0.f);
      }
    }
  if (do_bias) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float v = bsum[i] + __shfl_xor(bsum[i], 32, 64);   // the two row parities
      const int n = n0 + i * 32 + l31;
      if (lh == 0 && n < a.N) {
        float* bo = a.S > 1 ? a.bias_partial + (size_t)blockIdx.y * a.N : a.bias;
        bo[n] = v + (acc_c ? bo[n] : 0.f);
      }
    }
  }
}

// The same product with the operands staged through LDS: 16-byte coalesced global loads (a quarter of the load
// instructions and of the texture-addresser time of the register version, which issues one 4-byte load per lane and
// MFMA operand), each element fetched once per workgroup and read by the waves that need it from LDS in the layout it
// has in memory -- [row m][column]: lane (column = lane & 31, m parity = lane >> 5) is exactly the 32x32x2 operand
// order, consecutive lanes hit consecutive banks.  Chunks of 32 rows, double-buffered.  Needs 16-byte aligned rows
// (lda, ldb multiples of 4, aligned bases); the register kernel takes the rest.
// MC = rows per chunk: 32 with one workgroup per CU, 16 when the reduction is split far enough for two or more (long
// reductions: the second workgroup of a CU hides what a single wave per SIMD cannot -- load latency, the LDS writes;
// measured on 32768 x 512 x 512 with scripts/dev/atb_lab.hip: 176 us at 256 workgroups x 32 rows, 138 at 512 x 16).
namespace atbl {
constexpr int NT = 256, BN = 128, BK = 128;
}  // namespace atbl

template <int MC>
__global__ __launch_bounds__(atbl::NT) void gemm_atb_lds_kernel(AtbArgs a) {
  using namespace atbl;
  constexpr int LDS_FLOATS = 2 * MC * (BN + BK);
  constexpr int P = MC / 8;   // rows a thread stages per chunk and operand
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
  const int tiles_k = (a.K + BK - 1) / BK;
  const int tn = blockIdx.x / tiles_k, tk = blockIdx.x - tn * tiles_k;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int nw = (wave >> 1) * 64, kw = (wave & 1) * 64;      // this wave's 64 x 64 inside the tile
  const int n_base = tn * BN, k_base = tk * BK;
  const int chunk = (((a.M + a.S - 1) / a.S) + MC - 1) / MC * MC;
  const int ms = blockIdx.y * chunk, me = min(a.M, ms + chunk);
  f32x16t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // staging: thread (row r8 = tid / 32, 16-byte piece c4 = tid % 32) moves rows r8 + 8 p, p < P, of both operands
  const int r8 = tid >> 5, c4 = (tid & 31) * 4;
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 ga[P], gb[P];
  // column sums of A (the bias gradient): every row passes through the staging registers exactly once, so the threads
  // add what they write to LDS (per thread: rows r8 + 8 j of four columns); the eight row groups meet in LDS at the end
  const bool do_bias = a.bias_partial != nullptr && tk == 0;   // uniform over the workgroup
  f4 bs = {0.f, 0.f, 0.f, 0.f};
  const bool tile_full = n_base + BN <= a.N && k_base + BK <= a.K;   // uniform over the workgroup
  // row segments (the applications of one network): a chunk never straddles two (seg_rows is a multiple of 32), and
  // the segment is looked up again only when the walk leaves it
  const float* A0 = a.A; const float* B0 = a.B;
  // Operand transform of the fused train-mode layer (kernels.h AtbArgs): a thread stages the same four columns of every
  // row, so its coefficients are registers, re-read only when the walk enters another segment (= application).
  f4 cbs = {1.f, 1.f, 1.f, 1.f}, cbt = {0.f, 0.f, 0.f, 0.f};
  const float b_slope = a.b_mode ? a.b_slope[0] : 0.f;
  const int b_cols = min(4, max(0, a.K - (k_base + c4)));
  auto load_coefs = [&](int sg) {
    if (a.b_mode && b_cols > 0) {
      const float* c = a.Bs_seg[sg] + k_base + c4;
      for (int e = 0; e < b_cols; ++e) { cbs[e] = c[e]; cbt[e] = c[a.K + e]; }
    }
  };
  if (a.n_seg == 0) load_coefs(0);
  bool rv[P];
  int mrel = 0, seg_end = a.n_seg > 0 ? 0 : 0x7fffffff;
  auto gload = [&](int m0) {
    if (m0 >= seg_end) {
      const int sg = min(m0 / a.seg_rows, a.n_seg - 1);
      A0 = a.A_seg[sg]; B0 = a.B_seg[sg]; mrel = sg * a.seg_rows;
      load_coefs(sg);
      seg_end = sg + 1 < a.n_seg ? mrel + a.seg_rows : 0x7fffffff;
    }
    if (tile_full && m0 + MC <= me) {   // interior chunk (all but the edges): straight loads, no per-lane branches
      const float* pa = A0 + (size_t)(m0 - mrel + r8) * a.lda + n_base + c4;
      const float* pb = B0 + (size_t)(m0 - mrel + r8) * a.ldb + k_base + c4;
#pragma unroll
      for (int p = 0; p < P; ++p) {
        ga[p] = *reinterpret_cast<const f4*>(pa + (size_t)(8 * p) * a.lda);
        gb[p] = *reinterpret_cast<const f4*>(pb + (size_t)(8 * p) * a.ldb);
        rv[p] = true;
      }
      return;
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int mm = m0 + r8 + 8 * p;
      const bool row_ok = mm < me;
      f4 va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
      if (row_ok) {
        const float* pa = A0 + (size_t)(mm - mrel) * a.lda + n_base + c4;
        const float* pb = B0 + (size_t)(mm - mrel) * a.ldb + k_base + c4;
        if (n_base + c4 + 4 <= a.N) va = *reinterpret_cast<const f4*>(pa);
        else
          for (int e = 0; e < 4; ++e) if (n_base + c4 + e < a.N) va[e] = pa[e];
        if (k_base + c4 + 4 <= a.K) vb = *reinterpret_cast<const f4*>(pb);
        else
          for (int e = 0; e < 4; ++e) if (k_base + c4 + e < a.K) vb[e] = pb[e];
      }
      ga[p] = va; gb[p] = vb; rv[p] = row_ok;
    }
  };
  auto lwrite = [&](float* st) {
#pragma unroll
    for (int p = 0; p < P; ++p) {
      if (a.b_mode && rv[p]) {   // a_{l-1} = PReLU(s y + t)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (e < b_cols) {
            const float y = cbs[e] * gb[p][e] + cbt[e];
            gb[p][e] = y > 0.f ? y : b_slope * y;
          }
      }
      *reinterpret_cast<f4*>(st + (r8 + 8 * p) * BN + c4) = ga[p];
      *reinterpret_cast<f4*>(st + MC * BN + (r8 + 8 * p) * BK + c4) = gb[p];
      if (do_bias) bs += ga[p];   // here, not at the load: the loads have landed by now
    }
  };
  constexpr int STAGE = MC * (BN + BK);
  gload(ms);
  lwrite(lds);
  __syncthreads();
  int buf = 0;
  for (int m = ms; m < me; m += MC) {
    const bool more = m + MC < me;
    if (more) gload(m + MC);
    const float* sA = lds + buf * STAGE + nw + l31;
    const float* sB = lds + buf * STAGE + MC * BN + kw + l31;
    // One wave per SIMD: nothing else hides the LDS latency, so the operands of row pair q + 1 are read before the
    // four MFMAs of row pair q (order pinned: left alone the compiler reads each pair right before its use).
    float a0 = sA[lh * BN], a1 = sA[lh * BN + 32];
    float b0 = sB[lh * BK], b1 = sB[lh * BK + 32];
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
    for (int q = 0; q < MC / 2; ++q) {
      const int row = (q + 1 < MC / 2 ? 2 * (q + 1) : 0) + lh;
      const float na0 = sA[row * BN], na1 = sA[row * BN + 32];
      const float nb0 = sB[row * BK], nb1 = sB[row * BK + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // the two ds_read2 of the next row pair
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);   // this row pair's MFMAs
      a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
    }
    if (more) lwrite(lds + (buf ^ 1) * STAGE);
    __syncthreads();
    buf ^= 1;
  }
  const int n0 = n_base + nw, k0 = k_base + kw;
  float* out = a.S > 1 ? a.partial + (size_t)blockIdx.y * a.N * a.K : a.C;
  const int ldo = a.S > 1 ? a.K : a.ldc;
  const bool acc_c = a.S == 1 && a.accumulate;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = k0 + j * 32 + l31;
      if (k >= a.K) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (n < a.N) out[(size_t)n * ldo + k] = acc[i][j][r] + (acc_c ? out[(size_t)n * ldo + k] : 0.f);
      }
    }
  if (do_bias) {   // (the loop left through a barrier: the LDS stages are free)
    *reinterpret_cast<f4*>(lds + r8 * BN + c4) = bs;
    __syncthreads();
    if (tid < BN && n_base + tid < a.N) {
      float v = 0.f;
#pragma unroll
      for (int g = 0; g < 8; ++g) v += lds[g * BN + tid];
      float* bo = a.S > 1 ? a.bias_partial + (size_t)blockIdx.y * a.N : a.bias;
      bo[n_base + tid] = v + (acc_c ? bo[n_base + tid] : 0.f);
    }
  }
}

// The interior of the same product as its own kernel (round 4): whole 128 x 128 tiles, every workgroup the same number
// of whole MC-row chunks inside ONE row segment, plain operands (no transform).  That is what the hidden layers of the
// update networks give (32768 x 512 x 512 over four applications), and for it the general kernel above carries its edge
// handling, segment walk and transform switches through the loop as branches and scalar reloads: 165 us where a stripped
// copy of the loop measured 138 (scripts/dev/atb_lab.hip).  Same staging, same operand order, same split -- the same bits.
struct AtbFastArgs {
  const float* A; const float* B; int lda, ldb;      // (segment bases are resolved per workgroup on the host side of the loop)
  const float* A_seg[ATB_MAX_SEG]; const float* B_seg[ATB_MAX_SEG];
  int n_seg, seg_rows;
  float* out; int ldo;            // partial tiles [S][N][K] (ldo = K) or C itself (S == 1)
  float* bias_out;                // [S][N] partial column sums, the bias itself (S == 1), or nullptr
  int N, K, tiles_k, chunk_rows;  // rows per workgroup (a multiple of MC)
  int acc_c;                      // S == 1 and accumulate
};

template <int MC>
__global__ __launch_bounds__(atbl::NT) void gemm_atb_fast_kernel(AtbFastArgs a) {
  using namespace atbl;
  constexpr int LDS_FLOATS = 2 * MC * (BN + BK);
  constexpr int P = MC / 8;
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
  const int tn = blockIdx.x / a.tiles_k, tk = blockIdx.x - tn * a.tiles_k;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int nw = (wave >> 1) * 64, kw = (wave & 1) * 64;
  const int n_base = tn * BN, k_base = tk * BK;
  const int ms = blockIdx.y * a.chunk_rows;
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int r8 = tid >> 5, c4 = (tid & 31) * 4;
  // this workgroup's rows lie in one segment: two row pointers, advanced by MC rows per chunk
  const float* pa; const float* pb;
  {
    const int sg = a.n_seg > 0 ? ms / a.seg_rows : 0;
    const int mrel = a.n_seg > 0 ? ms - sg * a.seg_rows : ms;
    const float* A0 = a.n_seg > 0 ? a.A_seg[sg] : a.A;
    const float* B0 = a.n_seg > 0 ? a.B_seg[sg] : a.B;
    pa = A0 + (size_t)(mrel + r8) * a.lda + n_base + c4;
    pb = B0 + (size_t)(mrel + r8) * a.ldb + k_base + c4;
  }
  const size_t step_a = (size_t)MC * a.lda, step_b = (size_t)MC * a.ldb;
  const size_t row8_a = (size_t)8 * a.lda, row8_b = (size_t)8 * a.ldb;
  f32x16t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  f4 ga[P], gb[P];
  const bool do_bias = a.bias_out != nullptr && tk == 0;   // uniform over the workgroup
  f4 bs = {0.f, 0.f, 0.f, 0.f};
  auto gload = [&]() {
#pragma unroll
    for (int p = 0; p < P; ++p) {
      ga[p] = *reinterpret_cast<const f4*>(pa + p * row8_a);
      gb[p] = *reinterpret_cast<const f4*>(pb + p * row8_b);
    }
    pa += step_a; pb += step_b;
  };
  auto lwrite = [&](float* st) {
#pragma unroll
    for (int p = 0; p < P; ++p) {
      *reinterpret_cast<f4*>(st + (r8 + 8 * p) * BN + c4) = ga[p];
      *reinterpret_cast<f4*>(st + MC * BN + (r8 + 8 * p) * BK + c4) = gb[p];
      if (do_bias) bs += ga[p];
    }
  };
  constexpr int STAGE = MC * (BN + BK);
  gload();
  lwrite(lds);
  __syncthreads();
  int buf = 0;
  const int n_chunks = a.chunk_rows / MC;
  for (int ch = 0; ch < n_chunks; ++ch) {
    const bool more = ch + 1 < n_chunks;
    if (more) gload();
    const float* sA = lds + buf * STAGE + nw + l31;
    const float* sB = lds + buf * STAGE + MC * BN + kw + l31;
    float a0 = sA[lh * BN], a1 = sA[lh * BN + 32];
    float b0 = sB[lh * BK], b1 = sB[lh * BK + 32];
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
    for (int q = 0; q < MC / 2; ++q) {
      const int row = (q + 1 < MC / 2 ? 2 * (q + 1) : 0) + lh;
      const float na0 = sA[row * BN], na1 = sA[row * BN + 32];
      const float nb0 = sB[row * BK], nb1 = sB[row * BK + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
    }
    if (more) lwrite(lds + (buf ^ 1) * STAGE);
    __syncthreads();
    buf ^= 1;
  }
  const int n0 = n_base + nw, k0 = k_base + kw;
  float* out = a.out + (gridDim.y > 1 ? (size_t)blockIdx.y * a.N * a.K : 0);   // the split's partial tile, or C itself
  const int ldo = a.ldo;
  const bool acc_c = a.acc_c != 0;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = k0 + j * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        out[(size_t)n * ldo + k] = acc[i][j][r] + (acc_c ? out[(size_t)n * ldo + k] : 0.f);
      }
    }
  if (do_bias) {
    *reinterpret_cast<f4*>(lds + r8 * BN + c4) = bs;
    __syncthreads();
    if (tid < BN) {
      float v = 0.f;
#pragma unroll
      for (int g = 0; g < 8; ++g) v += lds[g * BN + tid];
      float* bo = a.bias_out + (gridDim.y > 1 ? (size_t)blockIdx.y * a.N : 0);
      bo[n_base + tid] = v + (acc_c ? bo[n_base + tid] : 0.f);
    }
  }
}

// C[i] = sum_s partial[s][i] in split order; the same for the bias partials
__global__ void atb_reduce_kernel(AtbArgs a) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nk = (size_t)a.N * a.K;
  if (i < nk) {
    float v = 0.f;
    int s = 0;
    for (; s + 8 <= a.S; s += 8) {   // eight loads in flight, added in split order
      float t[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) t[j] = a.partial[(size_t)(s + j) * nk + i];
#pragma unroll
      for (int j = 0; j < 8; ++j) v += t[j];
    }
    for (; s < a.S; ++s) v += a.partial[(size_t)s * nk + i];
    const size_t n = i / a.K, k = i - n * a.K;
    a.C[n * a.ldc + k] = v + (a.accumulate ? a.C[n * a.ldc + k] : 0.f);
  }
  if (a.bias_partial && i < (size_t)a.N) {
    float v = 0.f;
    for (int s = 0; s < a.S; ++s) v += a.bias_partial[(size_t)s * a.N + i];
    a.bias[i] = v + (a.accumulate ? a.bias[i] : 0.f);
  }
}

// Long reductions (>= 768 rows per workgroup at twice the target) split twice as far and stage 16-row chunks, so that
// two workgroups share a CU (scripts/dev/bench_atb.py: 32768 x 512 x 512 216 -> 172 us, 32768 x 512 x 296 254 -> 195,
// 16384 x 512 x 512 and below unchanged); "atb_target" / "atb_chunk" (dev options) override both.
static bool atb_long(int M, int tiles) { return (long)M * tiles >= (long)512 * 768; }
static int atb_chunk_rows(int M, int N, int K) {
  const int tiles = ((N + atb::BN - 1) / atb::BN) * ((K + atb::BK - 1) / atb::BK);
  if (options().atb_chunk == 16 || options().atb_chunk == 32) return options().atb_chunk;
  return atb_long(M, tiles) ? 16 : 32;
}
int atb_splits(int M, int N, int K) {
  const int tiles = ((N + atb::BN - 1) / atb::BN) * ((K + atb::BK - 1) / atb::BK);
  const int target = options().atb_target > 0 ? options().atb_target : (atb_long(M, tiles) ? 512 : 256);
  int s = (target + tiles - 1) / tiles;
  s = std::min(s, std::max(1, M / 64));
  return std::max(1, std::min(s, 256));
}

size_t atb_workspace_floats(int M, int N, int K) {
  const int s = atb_splits(M, N, K);
  return s > 1 ? (size_t)s * ((size_t)N * K + N) : 0;
}

size_t atb_workspace_floats_max(int M, std::initializer_list<std::pair<int, int>> products) {
  size_t m = 0;
  for (const auto& nk : products)
    if (nk.first > 0 && nk.second > 0) m = std::max(m, atb_workspace_floats(M, nk.first, nk.second));
  return m;
}

// `workspace_floats` is what the caller carved: the split count grows as the tile count shrinks, so a workspace sized
// for ONE product shape is not automatically large enough for a smaller one.  Callers size it over every
// product they launch (atb_workspace_floats_max); should one ever come up short, the split count is reduced to what
// fits instead of writing past the buffer.
hipError_t launch_gemm_atb(AtbArgs a, float* workspace, size_t workspace_floats, hipStream_t stream) {
  a.S = atb_splits(a.M, a.N, a.K);
  if (a.S > 1) {
    const size_t per_split = (size_t)a.N * a.K + a.N;
    const size_t fit = workspace ? workspace_floats / per_split : 0;
    if (fit < (size_t)a.S) a.S = fit >= 2 ? (int)fit : 1;
  }
  a.partial = nullptr;
  a.bias_partial = a.bias;   // S == 1: the kernel writes the bias directly (pointer doubles as the flag)
  if (a.S > 1) {
    a.partial = workspace;
    a.bias_partial = a.bias ? workspace + (size_t)a.S * a.N * a.K : nullptr;
  }
  const int tiles = ((a.N + atb::BN - 1) / atb::BN) * ((a.K + atb::BK - 1) / atb::BK);
  bool aligned = a.lda % 4 == 0 && a.ldb % 4 == 0;
  if (a.n_seg > 0) {
    for (int s = 0; s < a.n_seg; ++s) aligned = aligned && ((uintptr_t)a.A_seg[s] & 15) == 0 && ((uintptr_t)a.B_seg[s] & 15) == 0;
    if (!aligned || a.seg_rows % 32 != 0 || a.n_seg > ATB_MAX_SEG) return hipErrorInvalidValue;   // callers check first
  } else {
    aligned = aligned && ((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.B & 15) == 0;
  }
  if (a.b_mode && !aligned) return hipErrorInvalidValue;   // callers keep these operands aligned
  const int mc = atb_chunk_rows(a.M, a.N, a.K);
  // the interior kernel: whole tiles, equal whole-chunk row ranges that do not straddle a segment, plain operands
  const int rows_per_wg = a.S > 0 ? a.M / a.S : 0;
  const bool fast = options().atb_fast && aligned && !a.b_mode && a.N % atbl::BN == 0 && a.K % atbl::BK == 0 &&
                    a.M % a.S == 0 && rows_per_wg % mc == 0 && rows_per_wg > 0 &&
                    (a.n_seg == 0 || a.seg_rows % rows_per_wg == 0) &&
                    ((((a.M + a.S - 1) / a.S) + mc - 1) / mc * mc) == rows_per_wg;   // = the split the general kernel would use
  if (fast) {
    AtbFastArgs f{};
    f.A = a.A; f.B = a.B; f.lda = a.lda; f.ldb = a.ldb; f.n_seg = a.n_seg; f.seg_rows = a.seg_rows;
    for (int s = 0; s < a.n_seg; ++s) { f.A_seg[s] = a.A_seg[s]; f.B_seg[s] = a.B_seg[s]; }
    f.out = a.S > 1 ? a.partial : a.C; f.ldo = a.S > 1 ? a.K : a.ldc;
    f.bias_out = a.bias_partial; f.N = a.N; f.K = a.K; f.tiles_k = a.K / atbl::BK; f.chunk_rows = rows_per_wg;
    f.acc_c = a.S == 1 && a.accumulate;
    if (mc == 16) hipLaunchKernelGGL(gemm_atb_fast_kernel<16>, dim3(tiles, a.S), dim3(atbl::NT), 0, stream, f);
    else hipLaunchKernelGGL(gemm_atb_fast_kernel<32>, dim3(tiles, a.S), dim3(atbl::NT), 0, stream, f);
  } else if (aligned && mc == 16)
    hipLaunchKernelGGL(gemm_atb_lds_kernel<16>, dim3(tiles, a.S), dim3(atbl::NT), 0, stream, a);
  else if (aligned) hipLaunchKernelGGL(gemm_atb_lds_kernel<32>, dim3(tiles, a.S), dim3(atbl::NT), 0, stream, a);
  else hipLaunchKernelGGL(gemm_atb_kernel, dim3(tiles, a.S), dim3(atb::NT), 0, stream, a);
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
  lstm_cell_bwd_elem(a, idx, a.dh_in ? a.dh_in[idx] : 0.f);
}

// ---------------------------------------------------------------------------------------------------------------
// Elementwise pieces of the training step
// ---------------------------------------------------------------------------------------------------------------
// out[t][c] = mean over the window of t (F consecutive rows) of in[.][c]; its adjoint is the same operator.
__global__ void window_mean_kernel(const float* in, int ld_in, float* out, int ld_out, int T, int F, int C) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= T * C) return;
  const int t = idx / C, c = idx - t * C;
  const int w0 = (t / F) * F;
  float s = 0.f;
  for (int f = 0; f < F; ++f) s += in[(size_t)(w0 + f) * ld_in + c];
  out[(size_t)t * ld_out + c] = s / (float)F;
}
hipError_t launch_window_mean(const float* in, int ld_in, float* out, int ld_out, int T, int F, int C, hipStream_t stream) {
  const long n = (long)T * C;
  hipLaunchKernelGGL(window_mean_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, in, ld_in, out, ld_out,
                     T, F, C);
  return hipGetLastError();
}

// out[r][c] = alpha * x[r][c] + beta * y[r][c] over a [rows][cols] block with row strides (out may alias x or y)
__global__ void axpby2d_kernel(int rows, int cols, float alpha, const float* x, int ldx, float beta, const float* y, int ldy,
                               float* out, int ldo) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)rows * cols) return;
  const int r = (int)(idx / cols), c = (int)(idx - (long)r * cols);
  const float xv = x ? x[(size_t)r * ldx + c] : 0.f;
  const float yv = y ? y[(size_t)r * ldy + c] : 0.f;
  out[(size_t)r * ldo + c] = alpha * xv + beta * yv;
}
hipError_t launch_axpby2d(int rows, int cols, float alpha, const float* x, int ldx, float beta, const float* y, int ldy,
                          float* out, int ldo, hipStream_t stream) {
  const long n = (long)rows * cols;
  hipLaunchKernelGGL(axpby2d_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, rows, cols, alpha, x, ldx,
                     beta, y, ldy, out, ldo);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// The bookkeeping of the LGD loop around the update networks, one launch each instead of a handful of axpby / mean
// launches (at the reference's training batch of 12 windows a step is launch-bound: every launch saved is ~5 us).
// Same operations in the same order as the separate kernels they replace.
// ---------------------------------------------------------------------------------------------------------------
// X[t] = [ x0[t] (d_in) | pose[t] (66) | shape[t] (10) | (the gradient columns are written by the body-model kernel) ]
__global__ void lgd_assemble_kernel(int T, int d_in, const float* x0, int ld0, const float* pose, const float* shape,
                                    float* X, int ldx) {
  const int w = d_in + 76;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)T * w) return;
  const int t = (int)(idx / w), c = (int)(idx - (long)t * w);
  float v;
  if (c < d_in) v = x0[(size_t)t * ld0 + c];
  else if (c < d_in + 66) v = pose[(size_t)t * 66 + (c - d_in)];
  else v = shape[(size_t)t * 10 + (c - d_in - 66)];
  X[(size_t)t * ldx + c] = v;
}
hipError_t launch_lgd_assemble(int T, int d_in, const float* x0, int ld0, const float* pose, const float* shape, float* X,
                               int ldx, hipStream_t stream) {
  const long n = (long)T * (d_in + 76);
  hipLaunchKernelGGL(lgd_assemble_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, T, d_in, x0, ld0, pose,
                     shape, X, ldx);
  return hipGetLastError();
}

// pose_next = pose + step * d_pose;  shape_next = shape + step * (shape_avg ? window mean of d_shape : d_shape).
// One workgroup per window of F frames (reference models.py:529-535, 588-600).
// (grid (B, 1 + LGD_POSE_PARTS): y = 0 the window's shape columns -- the only part that needs the whole window --, y >= 1 a
// slice of its F x 66 pose entries; one workgroup per window was a serial walk of ten loads deep, 12-18 us per launch at the
// reference's 12 windows.  Same arithmetic per element.)
constexpr int LGD_POSE_PARTS = 4;
__global__ __launch_bounds__(256) void lgd_update_kernel(int F, float step, int shape_avg, const float* pose,
                                                         const float* d_pose, const float* shape, const float* d_shape,
                                                         float* pose_next, float* shape_next) {
  __shared__ float mean[10];
  const size_t t0 = (size_t)blockIdx.x * F;
  if (blockIdx.y > 0) {
    const int n = F * 66, per = (n + LGD_POSE_PARTS - 1) / LGD_POSE_PARTS;
    const int lo = (blockIdx.y - 1) * per, hi = min(n, lo + per);
    for (int i = lo + threadIdx.x; i < hi; i += 256) pose_next[t0 * 66 + i] = step * d_pose[t0 * 66 + i] + pose[t0 * 66 + i];
    return;
  }
  if (shape_avg && threadIdx.x < 10) {
    float s = 0.f;
    for (int f = 0; f < F; ++f) s += d_shape[(t0 + f) * 10 + threadIdx.x];
    mean[threadIdx.x] = s / (float)F;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < F * 10; i += 256) {
    const int k = i % 10;
    const float d = shape_avg ? mean[k] : d_shape[t0 * 10 + i];
    shape_next[t0 * 10 + i] = step * d + shape[t0 * 10 + i];
  }
}
hipError_t launch_lgd_update(int B, int F, float step, int shape_avg, const float* pose, const float* d_pose,
                             const float* shape, const float* d_shape, float* pose_next, float* shape_next,
                             hipStream_t stream) {
  hipLaunchKernelGGL(lgd_update_kernel, dim3(B, 1 + LGD_POSE_PARTS), dim3(256), 0, stream, F, step, shape_avg, pose, d_pose,
                     shape, d_shape, pose_next, shape_next);
  return hipGetLastError();
}

// Reverse sweep, entry i of the histories: the running cotangents of the estimates
//   Dp = [Dp +] d_pose_i + vp_i [+ g_theta_i / T],   Ds likewise           (loss terms, body-model VJP, the reference's
//   in-forward E_i.backward() deposit, models.py:576)
// and, for i > 0, the cotangents of the update networks' outputs of iteration i - 1, zero-padded to the GEMM grid:
//   dpad[:, :66] = step * Dp,   dspad[:, :10] = step * (shape_avg ? window mean of Ds : Ds)   (adjoint of the mean = mean)
// (the padding columns are written too, so the destinations may be uninitialised memory: the stash slots of
// empose_mlp_train_bwd_deferred)
__global__ __launch_bounds__(256) void lgd_cotangent_kernel(int F, float inv_T, int first, const float* d_pose,
                                                            const float* d_shape, const float* vp, const float* vs,
                                                            const float* g_theta, int ld_g, const float* g_beta, int ld_gb,
                                                            float* Dp, float* Ds, float step, int shape_avg, float* dpad,
                                                            float* dspad) {
  extern __shared__ float sds[];   // [F][10]
  const size_t t0 = (size_t)blockIdx.x * F;
  if (blockIdx.y > 0) {            // a slice of the window's F x 66 pose entries (grid as lgd_update_kernel)
    const int n = F * 66, per = (n + LGD_POSE_PARTS - 1) / LGD_POSE_PARTS;
    const int lo = (blockIdx.y - 1) * per, hi = min(n, lo + per);
    for (int i = lo + threadIdx.x; i < hi; i += 256) {
      const int f = i / 66, c = i - f * 66;
      const size_t t = t0 + f;
      float v = d_pose[t * 66 + c];
      if (!first) v = v + Dp[t * 66 + c];
      v = vp[t * 66 + c] + v;
      if (g_theta) v = inv_T * g_theta[t * ld_g + c] + v;
      Dp[t * 66 + c] = v;
      if (dpad) {
        dpad[t * 68 + c] = step * v;
        if (c >= 64) dpad[t * 68 + c + 2] = 0.f;   // the two padding columns
      }
    }
    return;
  }
  for (int i = threadIdx.x; i < F * 10; i += 256) {
    const int f = i / 10, k = i - f * 10;
    const size_t t = t0 + f;
    float v = d_shape[t * 10 + k];
    if (!first) v = v + Ds[t * 10 + k];
    v = vs[t * 10 + k] + v;
    if (g_beta) v = inv_T * g_beta[t * ld_gb + k] + v;
    Ds[t * 10 + k] = v;
    sds[f * 10 + k] = v;
  }
  if (!dspad) return;
  __syncthreads();
  for (int i = threadIdx.x; i < F * 10; i += 256) {
    const int f = i / 10, k = i - f * 10;
    float v;
    if (shape_avg) {
      float s = 0.f;
      for (int ff = 0; ff < F; ++ff) s += sds[ff * 10 + k];
      v = s / (float)F;
    } else {
      v = sds[i];
    }
    dspad[(t0 + f) * 12 + k] = step * v;
    if (k >= 8) dspad[(t0 + f) * 12 + k + 2] = 0.f;   // padding columns
  }
}
hipError_t launch_lgd_cotangent(int B, int F, int first, const float* d_pose, const float* d_shape, const float* vp,
                                const float* vs, const float* g_theta, int ld_g, const float* g_beta, int ld_gb, float* Dp,
                                float* Ds, float step, int shape_avg, float* dpad, float* dspad, hipStream_t stream) {
  const float inv_T = 1.f / (float)((long)B * F);
  hipLaunchKernelGGL(lgd_cotangent_kernel, dim3(B, 1 + LGD_POSE_PARTS), dim3(256), (size_t)F * 10 * sizeof(float), stream, F, inv_T, first,
                     d_pose, d_shape, vp, vs, g_theta, ld_g, g_beta, ld_gb, Dp, Ds, step, shape_avg, dpad, dspad);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Losses of IterativeErrorFeedback.backward and their cotangents (reference models.py:634-688, loss.py:13-41):
//   total = (w_pose sum_i L1(pose_i) + w_shape sum_i L1(shape_i) + w_fk (N+1) FK(joints_N) + w_rec sum_i REC_i) / (N+1)
//   L1: |hat - gt| averaged over the features, summed over the valid frames / len_b, averaged over the batch
//   FK / REC: sum of Euclidean norms per frame (joints; sensor positions + 9-vector orientations of the sensors fed to
//   the network), frames with a missing sensor dropped, summed over the valid frames / len_b, averaged over the batch.
// 16 lanes per (history entry, frame): lane 0 pose, 1 shape, 2..13 one sensor each, 14 the joints (last entry only).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lgd_losses_kernel(LossArgs a) {
  const int T = a.B * a.F;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long item = gid >> 4;
  const int part = (int)(gid & 15);
  const bool in_range = item < (long)a.N1 * T;
  const long it = in_range ? item : 0;
  const int i = (int)(it / T), t = (int)(it - (long)i * T);
  const int b = t / a.F, f = t - b * a.F;
  const int len = a.seq_lengths ? a.seq_lengths[b] : a.F;
  const bool live = in_range && f < len;
  bool frame_ok = true;
  if (a.masks)
    for (int m = 0; m < 12; ++m) frame_ok = frame_ok && a.masks[(size_t)t * 12 + m] != 0.f;
  const float inv = 1.f / ((float)len * (float)a.B);
  const float inv_n1 = 1.f / (float)a.N1;
  float l_pose = 0.f, l_shape = 0.f, l_rec = 0.f, l_fk = 0.f;
  const size_t row = (size_t)i * T + t;
  if (in_range && part == 0) {
    const float* h = a.pose_hist + row * 66;
    const float* g = a.pose_gt + (size_t)t * 66;
    float* d = a.d_pose + row * 66;
    const float k = live ? a.w_pose * inv_n1 * inv / 66.f : 0.f;
    float s = 0.f;
    for (int c = 0; c < 66; ++c) {
      const float df = h[c] - g[c];
      s += fabsf(df);
      d[c] = df > 0.f ? k : (df < 0.f ? -k : 0.f);
    }
    l_pose = live ? s / 66.f * inv : 0.f;
  } else if (in_range && part == 1) {
    const float* h = a.shape_hist + row * 10;
    const float* g = a.shape_gt + (size_t)b * 10;
    float* d = a.d_shape + row * 10;
    const float k = live ? a.w_shape * inv_n1 * inv / 10.f : 0.f;
    float s = 0.f;
    for (int c = 0; c < 10; ++c) {
      const float df = h[c] - g[c];
      s += fabsf(df);
      d[c] = df > 0.f ? k : (df < 0.f ? -k : 0.f);
    }
    l_shape = live ? s / 10.f * inv : 0.f;
  } else if (in_range && part >= 2 && part < 14) {
    const int m = part - 2;
    const int slot = a.used_slot[m];
    float* dp = a.d_pos + (row * 12 + m) * 3;
    float* dq = a.d_ori + (row * 12 + m) * 9;
    const bool on = live && frame_ok && slot >= 0;
    if (!on) {
      for (int c = 0; c < 3; ++c) dp[c] = 0.f;
      for (int c = 0; c < 9; ++c) dq[c] = 0.f;
    } else {
      const float* hp = a.pos_hist + (row * 12 + m) * 3;
      const float* hq = a.ori_hist + (row * 12 + m) * 9;
      const float* gp = a.x_in + (size_t)t * a.ldx + slot * 3;
      const float* gq = a.x_in + (size_t)t * a.ldx + a.n_markers * 3 + slot * 9;
      const float k = a.w_rec * inv_n1 * inv;
      float r[9], q = 0.f;
      for (int c = 0; c < 3; ++c) { r[c] = hp[c] - gp[c]; q += r[c] * r[c]; }
      float nrm = sqrtf(q);
      for (int c = 0; c < 3; ++c) dp[c] = k * r[c] / nrm;
      l_rec = nrm * inv;
      q = 0.f;
      for (int c = 0; c < 9; ++c) { r[c] = hq[c] - gq[c]; q += r[c] * r[c]; }
      nrm = sqrtf(q);
      for (int c = 0; c < 9; ++c) dq[c] = k * r[c] / nrm;
      l_rec += nrm * inv;
    }
  } else if (in_range && part == 14 && i == a.N1 - 1 && a.d_joints) {
    float* d = a.d_joints + (size_t)t * 66;
    const bool on = live && frame_ok && a.joints_gt != nullptr;
    if (!on) {
      for (int c = 0; c < 66; ++c) d[c] = 0.f;
    } else {
      const float* h = a.joints_final + (size_t)t * 66;
      const float* g = a.joints_gt + (size_t)t * 66;
      const float k = a.w_fk * inv;   // added N + 1 times, divided by N + 1
      float s = 0.f;
      for (int j = 0; j < 22; ++j) {
        const float r0 = h[j * 3] - g[j * 3], r1 = h[j * 3 + 1] - g[j * 3 + 1], r2 = h[j * 3 + 2] - g[j * 3 + 2];
        const float nrm = sqrtf(r0 * r0 + r1 * r1 + r2 * r2);
        d[j * 3] = k * r0 / nrm; d[j * 3 + 1] = k * r1 / nrm; d[j * 3 + 2] = k * r2 / nrm;
        s += nrm;
      }
      l_fk = s * inv;
    }
  }
  // per (entry, frame) sums over the 16 lanes -> partial[kind][item]
#pragma unroll
  for (int off = 8; off >= 1; off >>= 1) {
    l_pose += __shfl_xor(l_pose, off, 16);
    l_shape += __shfl_xor(l_shape, off, 16);
    l_rec += __shfl_xor(l_rec, off, 16);
    l_fk += __shfl_xor(l_fk, off, 16);
  }
  if (in_range && part == 0) {
    const size_t n = (size_t)a.N1 * T;
    a.partial[item] = l_pose; a.partial[n + item] = l_shape; a.partial[2 * n + item] = l_rec; a.partial[3 * n + item] = l_fk;
  }
}

// loss_vals = (pose, shape, reconstruction, fk, total) from the per-frame contributions, summed in index order
__global__ __launch_bounds__(1024) void lgd_losses_reduce_kernel(LossArgs a) {
  // the four terms side by side: 256 threads each, eight loads in flight per thread, double accumulation in a fixed order
  // (one term after the other with one load at a time took 45 us at 256 windows)
  __shared__ double red[4][256];
  const size_t n = (size_t)a.N1 * a.B * a.F;
  const int k = threadIdx.x >> 8, t = threadIdx.x & 255;
  const float* src = a.partial + (size_t)k * n;
  double s = 0.0;
  size_t i = t;
  for (; i + 7 * 256 < n; i += 8 * 256) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[i + u * 256];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += (double)v[u];
  }
  for (; i < n; i += 256) s += (double)src[i];
  red[k][t] = s;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if (t < off) red[k][t] += red[k][t + off];
    __syncthreads();
  }
  const double sums[4] = {red[0][0], red[1][0], red[2][0], red[3][0]};
  if (threadIdx.x == 0) {
    const double n1 = (double)a.N1;
    const double fk_sum = sums[3] * n1;   // the same FK term is added once per history entry
    a.loss_vals[0] = (float)(sums[0] / n1);
    a.loss_vals[1] = (float)(sums[1] / n1);
    a.loss_vals[2] = (float)(sums[2] / n1);
    a.loss_vals[3] = (float)(fk_sum / n1);
    a.loss_vals[4] = (float)((a.w_pose * sums[0] + a.w_fk * fk_sum + a.w_shape * sums[1] + a.w_rec * sums[2]) / n1);
  }
}

hipError_t launch_lgd_losses(const LossArgs& a, hipStream_t stream) {
  const long n = (long)a.N1 * a.B * a.F * 16;
  hipLaunchKernelGGL(lgd_losses_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a);
  hipLaunchKernelGGL(lgd_losses_reduce_kernel, dim3(1), dim3(1024), 0, stream, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Adam over a list of tensors in one launch (torch.optim.Adam semantics, amsgrad off, weight_decay 0):
//   m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// chunk c covers elements [off, off + ADAM_CHUNK) of tensor `tensor_of_chunk[c]`.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a) {
  const int ch = blockIdx.x;
  const int ti = a.chunk_tensor[ch];
  const long base = (long)a.chunk_offset[ch];
  const long n = a.sizes[ti];
  float* p = reinterpret_cast<float*>(a.params[ti]);
  const float* g = reinterpret_cast<const float*>(a.grads[ti]);
  float* m = reinterpret_cast<float*>(a.exp_avg[ti]);
  float* v = reinterpret_cast<float*>(a.exp_avg_sq[ti]);
  for (int k = threadIdx.x; k < ADAM_CHUNK; k += 256) {
    const long i = base + k;
    if (i >= n) break;
    const float gi = g[i];
    const float mi = a.beta1 * m[i] + (1.f - a.beta1) * gi;
    const float vi = a.beta2 * v[i] + (1.f - a.beta2) * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] -= a.step_size * mi / (sqrtf(vi) * a.inv_sqrt_bc2 + a.eps);
  }
}
hipError_t launch_adam(const AdamArgs& a, int n_chunks, hipStream_t stream) {
  hipLaunchKernelGGL(adam_kernel, dim3(n_chunks), dim3(256), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_lstm_cell_bwd(const LstmCellBwdArgs& a, hipStream_t stream) {
  const int n = a.B * a.H;
  hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, a);
  return hipGetLastError();
}

}  // namespace empose
