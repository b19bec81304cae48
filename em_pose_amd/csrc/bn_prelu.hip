// Train-mode BatchNorm1d + PReLU of the update MLPs' hidden layers (reference nn/layers.py:13-77 in training mode:
// Linear -> BatchNorm1d -> PReLU(one shared slope) -> Dropout(p = 0)), forward and backward as ONE kernel each.
// The training step of the reference's batch (384 rows) is bound by the NUMBER of its ~2000 small kernels; the PyTorch
// ops for this pair are about eleven of them per layer (statistics, transform, running-stat updates, PReLU, and their
// backward reductions), forty layers per step.
//
// A workgroup owns 32 columns and all M rows (1024 threads = 32 row groups x 32 columns, coalesced 128-byte rows);
// column statistics are two-pass (mean, then centred sum of squares), reduced over the row groups in LDS.
//   forward:  mean, biased var -> rstd; y = gamma * (x - mean) * rstd + beta; z = y > 0 ? y : a * y;
//             running_mean/var updated with momentum (unbiased variance, as torch.nn.BatchNorm1d does)
//   backward: dy = y > 0 ? dz : a * dz;  da += sum_{y <= 0} dz * y;  dbeta = sum dy;  dgamma = sum dy * xhat;
//             dx = gamma * rstd / M * (M * dy - dbeta - xhat * dgamma)
// The slope gradient is one partial sum per workgroup; the workgroup that finishes last adds them in index order.
#include "kernels.h"

namespace empose {

namespace bp {
constexpr int COLS = 32, RG = 32, NT = COLS * RG;   // 1024 threads: 12 rows per thread and pass at M = 384
}

__device__ __forceinline__ float bp_reduce_rows(float v, float* red, int rg, int c) {
  red[rg * bp::COLS + c] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int g = 0; g < bp::RG; ++g) s += red[g * bp::COLS + c];
  __syncthreads();
  return s;
}

// R = rows per thread (M <= 32 R): a thread's rows stay in registers from the first pass to the last, so every operand
// is read from memory ONCE (three dependent trips to L2 per call before: 7.5 / 11.3 us per 384 x 512 layer, forty layer
// applications per training step at the reference's batch).  Same sums in the same order as before.
template <int R>
__global__ __launch_bounds__(bp::NT) void bn_prelu_fwd_kernel(BnPreluArgs a) {
  using namespace bp;
  __shared__ float red[RG * COLS];
  const int c = threadIdx.x & (COLS - 1), rg = threadIdx.x / COLS;
  const int col = blockIdx.x * COLS + c;
  const bool ok = col < a.C;
  const int cc = ok ? col : a.C - 1;
  const float inv_m = 1.f / (float)a.M;
  const float g = a.gamma[cc], b = a.beta[cc], slope = a.slope[0];
  float xv[R];
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int m = rg + i * RG;
    xv[i] = m < a.M ? a.x[(size_t)m * a.ldx + cc] : 0.f;
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < R; ++i) s += xv[i];          // (rows past M add 0: the same bits as stopping at M)
  const float mean = bp_reduce_rows(s, red, rg, c) * inv_m;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const float d = xv[i] - mean;
    q += (rg + i * RG < a.M) ? d * d : 0.f;
  }
  const float var = bp_reduce_rows(q, red, rg, c) * inv_m;      // biased: what normalises the batch
  const float rstd = 1.f / sqrtf(var + a.eps);
  if (ok) {
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int m = rg + i * RG;
      if (m < a.M) {
        const float y = g * ((xv[i] - mean) * rstd) + b;
        a.z[(size_t)m * a.ldz + col] = y > 0.f ? y : slope * y;
      }
    }
    if (rg == 0) {
      a.save_mean[col] = mean;
      a.save_rstd[col] = rstd;
      if (a.running_mean) {
        const float unbiased = a.M > 1 ? var * (float)a.M / (float)(a.M - 1) : var;
        a.running_mean[col] = (1.f - a.momentum) * a.running_mean[col] + a.momentum * mean;
        a.running_var[col] = (1.f - a.momentum) * a.running_var[col] + a.momentum * unbiased;
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && a.num_batches_tracked) a.num_batches_tracked[0] += 1;
}

template <int R>
__global__ __launch_bounds__(bp::NT) void bn_prelu_bwd_kernel(BnPreluArgs a) {
  using namespace bp;
  __shared__ float red[RG * COLS];
  const int c = threadIdx.x & (COLS - 1), rg = threadIdx.x / COLS;
  const int col = blockIdx.x * COLS + c;
  const bool ok = col < a.C;
  const int cc = ok ? col : a.C - 1;
  const float mean = a.save_mean[cc], rstd = a.save_rstd[cc];
  const float g = a.gamma[cc], b = a.beta[cc], slope = a.slope[0];
  float xh[R], dy[R];
  float s_b = 0.f, s_g = 0.f, s_a = 0.f;
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int m = rg + i * RG;
    const bool in = m < a.M;
    const float x = in ? a.x[(size_t)m * a.ldx + cc] : mean;
    const float dz = in ? a.dz[(size_t)m * a.lddz + cc] : 0.f;
    xh[i] = (x - mean) * rstd;
    const float y = g * xh[i] + b;
    dy[i] = y > 0.f ? dz : slope * dz;
    s_b += dy[i];
    s_g += dy[i] * xh[i];
    s_a += y > 0.f ? 0.f : dz * y;
  }
  const float dbeta = bp_reduce_rows(s_b, red, rg, c);
  const float dgamma = bp_reduce_rows(s_g, red, rg, c);
  float da = bp_reduce_rows(ok ? s_a : 0.f, red, rg, c);
  // slope: sum over this workgroup's columns (wave 0 holds one column per lane pair of halves)
  if (rg == 0) {
    red[c] = da;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < COLS; ++i) t += red[i];
    a.dslope_partial[blockIdx.x] = t;
    // the workgroup that arrives last adds the partial sums in index order (deterministic) and re-arms the counter
    __threadfence();
    if (atomicAdd(a.counter, 1) == (int)gridDim.x - 1) {
      __threadfence();
      float total = 0.f;
      for (unsigned i = 0; i < gridDim.x; ++i) total += __hip_atomic_load(a.dslope_partial + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      a.dslope[0] = total + (a.accumulate ? a.dslope[0] : 0.f);
      a.counter[0] = 0;
    }
  }
  if (!ok) return;
  const float k = g * rstd / (float)a.M;
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int m = rg + i * RG;
    if (m < a.M) a.dx[(size_t)m * a.lddx + col] = k * ((float)a.M * dy[i] - dbeta - xh[i] * dgamma);
  }
  if (rg == 0) {
    a.dgamma[col] = dgamma + (a.accumulate ? a.dgamma[col] : 0.f);
    a.dbeta[col] = dbeta + (a.accumulate ? a.dbeta[col] : 0.f);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Large batches (M > BN_SINGLE_PASS_ROWS: training with hundreds of windows per GPU): a workgroup per 32 columns and
// ALL rows leaves a 512-column layer on 16 workgroups (144 us per backward call at M = 8192).  Here the rows are split
// over workgroups too: pass 1 writes per-(row split, column) partial sums to a caller-provided workspace, pass 2 adds
// them in split order (deterministic), pass 3 transforms the rows.  The variance uses sums shifted by the column's first
// element, which keeps the one-pass form accurate.
// ---------------------------------------------------------------------------------------------------------------
namespace bpl {
constexpr int COLS = 64, RG = 4, NT = COLS * RG;   // 256 threads: 4 row groups x 64 columns
constexpr int ROWS = 128;                          // rows per workgroup
}

__device__ __forceinline__ float bpl_reduce(float v, float* red, int rg, int c) {
  red[rg * bpl::COLS + c] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int g = 0; g < bpl::RG; ++g) s += red[g * bpl::COLS + c];
  __syncthreads();
  return s;
}

// pass 1 forward: part[split][0][col] = sum (x - k), part[split][1][col] = sum (x - k)^2, k = x[0][col]
__global__ __launch_bounds__(bpl::NT) void bn_stats_fwd_kernel(BnPreluArgs a) {
  using namespace bpl;
  __shared__ float red[RG * COLS];
  const int c = threadIdx.x & (COLS - 1), rg = threadIdx.x / COLS;
  const int col = blockIdx.x * COLS + c;
  const int cc = col < a.C ? col : a.C - 1;
  const int m0 = blockIdx.y * ROWS, m1 = min(a.M, m0 + ROWS);
  const float k = a.x[cc];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll 4
  for (int m = m0 + rg; m < m1; m += RG) { const float d = a.x[(size_t)m * a.ldx + cc] - k; s1 += d; s2 += d * d; }
  s1 = bpl_reduce(s1, red, rg, c);
  s2 = bpl_reduce(s2, red, rg, c);
  if (rg == 0 && col < a.C) {
    float* p = a.workspace + (size_t)blockIdx.y * 3 * a.C;
    p[col] = s1; p[a.C + col] = s2;
  }
}

// pass 2 forward: a workgroup per 64 columns; 16 row groups each add a strided quarter of the partial sums, the row
// groups meet in LDS in a fixed order -> mean, rstd, running statistics
constexpr int BN_CG = 16;   // split groups of the combine kernels
__global__ __launch_bounds__(64 * BN_CG) void bn_combine_fwd_kernel(BnPreluArgs a, int n_split) {
  __shared__ float red[2][BN_CG][64];
  const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + c;
  const int cc = col < a.C ? col : a.C - 1;
  float s1 = 0.f, s2 = 0.f;
  for (int s = rg; s < n_split; s += BN_CG) { const float* p = a.workspace + (size_t)s * 3 * a.C; s1 += p[cc]; s2 += p[a.C + cc]; }
  red[0][rg][c] = s1; red[1][rg][c] = s2;
  __syncthreads();
  if (rg != 0 || col >= a.C) return;
  s1 = 0.f; s2 = 0.f;
#pragma unroll
  for (int g = 0; g < BN_CG; ++g) { s1 += red[0][g][c]; s2 += red[1][g][c]; }
  const float inv_m = 1.f / (float)a.M;
  const float d = s1 * inv_m;
  const float mean = a.x[col] + d;
  const float var = fmaxf(s2 * inv_m - d * d, 0.f);
  a.save_mean[col] = mean;
  a.save_rstd[col] = 1.f / sqrtf(var + a.eps);
  if (a.running_mean) {
    const float unbiased = a.M > 1 ? var * (float)a.M / (float)(a.M - 1) : var;
    a.running_mean[col] = (1.f - a.momentum) * a.running_mean[col] + a.momentum * mean;
    a.running_var[col] = (1.f - a.momentum) * a.running_var[col] + a.momentum * unbiased;
  }
  if (col == 0 && a.num_batches_tracked) a.num_batches_tracked[0] += 1;
}

// pass 3 forward: the transform
__global__ __launch_bounds__(bpl::NT) void bn_apply_fwd_kernel(BnPreluArgs a) {
  using namespace bpl;
  const int c = threadIdx.x & (COLS - 1), rg = threadIdx.x / COLS;
  const int col = blockIdx.x * COLS + c;
  if (col >= a.C) return;
  const float mean = a.save_mean[col], rstd = a.save_rstd[col];
  const float g = a.gamma[col], b = a.beta[col], slope = a.slope[0];
  const int m0 = blockIdx.y * ROWS, m1 = min(a.M, m0 + ROWS);
#pragma unroll 4
  for (int m = m0 + rg; m < m1; m += RG) {
    const float y = g * ((a.x[(size_t)m * a.ldx + col] - mean) * rstd) + b;
    a.z[(size_t)m * a.ldz + col] = y > 0.f ? y : slope * y;
  }
}

// pass 1 backward: part[split][0..2][col] = sum dy, sum dy * xhat, sum_{y <= 0} dz * y
__global__ __launch_bounds__(bpl::NT) void bn_stats_bwd_kernel(BnPreluArgs a) {
  using namespace bpl;
  __shared__ float red[RG * COLS];
  const int c = threadIdx.x & (COLS - 1), rg = threadIdx.x / COLS;
  const int col = blockIdx.x * COLS + c;
  const int cc = col < a.C ? col : a.C - 1;
  const int m0 = blockIdx.y * ROWS, m1 = min(a.M, m0 + ROWS);
  const float mean = a.save_mean[cc], rstd = a.save_rstd[cc];
  const float g = a.gamma[cc], b = a.beta[cc], slope = a.slope[0];
  float s_b = 0.f, s_g = 0.f, s_a = 0.f;
#pragma unroll 4
  for (int m = m0 + rg; m < m1; m += RG) {
    const float xh = (a.x[(size_t)m * a.ldx + cc] - mean) * rstd;
    const float y = g * xh + b;
    const float dz = a.dz[(size_t)m * a.lddz + cc];
    const float dy = y > 0.f ? dz : slope * dz;
    s_b += dy;
    s_g += dy * xh;
    s_a += y > 0.f ? 0.f : dz * y;
  }
  s_b = bpl_reduce(s_b, red, rg, c);
  s_g = bpl_reduce(s_g, red, rg, c);
  s_a = bpl_reduce(s_a, red, rg, c);
  if (rg == 0 && col < a.C) {
    float* p = a.workspace + (size_t)blockIdx.y * 3 * a.C;
    p[col] = s_b; p[a.C + col] = s_g; p[2 * a.C + col] = s_a;
  }
}

// pass 2 backward: a workgroup per 64 columns, split groups as in the forward; writes this call's (dbeta, dgamma)
// behind the partial sums for pass 3, the parameter gradients, and the workgroup's share of the slope gradient
__global__ __launch_bounds__(64 * BN_CG) void bn_combine_bwd_kernel(BnPreluArgs a, int n_split) {
  __shared__ float red[3][BN_CG][64];
  const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + c;
  const int cc = col < a.C ? col : a.C - 1;
  float s_b = 0.f, s_g = 0.f, s_a = 0.f;
  for (int s = rg; s < n_split; s += BN_CG) {
    const float* p = a.workspace + (size_t)s * 3 * a.C;
    s_b += p[cc]; s_g += p[a.C + cc]; s_a += p[2 * a.C + cc];
  }
  red[0][rg][c] = s_b; red[1][rg][c] = s_g; red[2][rg][c] = col < a.C ? s_a : 0.f;
  __syncthreads();
  if (rg != 0) return;
  s_b = 0.f; s_g = 0.f; s_a = 0.f;
#pragma unroll
  for (int g = 0; g < BN_CG; ++g) { s_b += red[0][g][c]; s_g += red[1][g][c]; s_a += red[2][g][c]; }
  if (col < a.C) {
    float* fin = a.workspace + (size_t)n_split * 3 * a.C;
    fin[col] = s_b; fin[a.C + col] = s_g;
    a.dgamma[col] = s_g + (a.accumulate ? a.dgamma[col] : 0.f);
    a.dbeta[col] = s_b + (a.accumulate ? a.dbeta[col] : 0.f);
  }
  // slope: the 64 columns of this workgroup in lane order (one wave)
  float t = s_a;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) t += __shfl_xor(t, off, 64);
  if (c == 0) a.dslope_partial[blockIdx.x] = t;
}

// pass 3 backward
__global__ __launch_bounds__(bpl::NT) void bn_apply_bwd_kernel(BnPreluArgs a, int n_split) {
  using namespace bpl;
  const int c = threadIdx.x & (COLS - 1), rg = threadIdx.x / COLS;
  const int col = blockIdx.x * COLS + c;
  if (col >= a.C) return;
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {   // slope gradient: the combine workgroups' shares in order
    float t = 0.f;
    for (unsigned i = 0; i < gridDim.x; ++i) t += a.dslope_partial[i];
    a.dslope[0] = t + (a.accumulate ? a.dslope[0] : 0.f);
  }
  const float* fin = a.workspace + (size_t)n_split * 3 * a.C;
  const float dbeta = fin[col], dgamma = fin[a.C + col];
  const float mean = a.save_mean[col], rstd = a.save_rstd[col];
  const float g = a.gamma[col], b = a.beta[col], slope = a.slope[0];
  const float k = g * rstd / (float)a.M;
  const int m0 = blockIdx.y * ROWS, m1 = min(a.M, m0 + ROWS);
#pragma unroll 4
  for (int m = m0 + rg; m < m1; m += RG) {
    const float xh = (a.x[(size_t)m * a.ldx + col] - mean) * rstd;
    const float y = g * xh + b;
    const float dz = a.dz[(size_t)m * a.lddz + col];
    const float dy = y > 0.f ? dz : slope * dz;
    a.dx[(size_t)m * a.lddx + col] = k * ((float)a.M * dy - dbeta - xh * dgamma);
  }
}

size_t bn_prelu_workspace_floats(int M, int C) {
  if (M <= BN_SINGLE_PASS_ROWS) return 0;
  return (size_t)((M + bpl::ROWS - 1) / bpl::ROWS + 1) * 3 * C;   // partial sums per row split + the combined sums
}

hipError_t launch_bn_prelu(const BnPreluArgs& a, bool backward, hipStream_t stream) {
  if (a.M > BN_SINGLE_PASS_ROWS) {
    const int n_split = (a.M + bpl::ROWS - 1) / bpl::ROWS;
    dim3 grid((a.C + bpl::COLS - 1) / bpl::COLS, n_split);
    if (backward) {
      hipLaunchKernelGGL(bn_stats_bwd_kernel, grid, dim3(bpl::NT), 0, stream, a);
      hipLaunchKernelGGL(bn_combine_bwd_kernel, dim3(grid.x), dim3(64 * BN_CG), 0, stream, a, n_split);
      hipLaunchKernelGGL(bn_apply_bwd_kernel, grid, dim3(bpl::NT), 0, stream, a, n_split);
    } else {
      hipLaunchKernelGGL(bn_stats_fwd_kernel, grid, dim3(bpl::NT), 0, stream, a);
      hipLaunchKernelGGL(bn_combine_fwd_kernel, dim3(grid.x), dim3(64 * BN_CG), 0, stream, a, n_split);
      hipLaunchKernelGGL(bn_apply_fwd_kernel, grid, dim3(bpl::NT), 0, stream, a);
    }
    return hipGetLastError();
  }
  const int blocks = (a.C + bp::COLS - 1) / bp::COLS;
  const int r = (a.M + bp::RG - 1) / bp::RG;   // rows per thread, <= 32
#define BP_LAUNCH(R)                                                                                         \
  do {                                                                                                       \
    if (backward) hipLaunchKernelGGL(bn_prelu_bwd_kernel<R>, dim3(blocks), dim3(bp::NT), 0, stream, a);      \
    else hipLaunchKernelGGL(bn_prelu_fwd_kernel<R>, dim3(blocks), dim3(bp::NT), 0, stream, a);               \
  } while (0)
  if (r <= 4) BP_LAUNCH(4);
  else if (r <= 8) BP_LAUNCH(8);
  else if (r <= 12) BP_LAUNCH(12);
  else if (r <= 16) BP_LAUNCH(16);
  else if (r <= 24) BP_LAUNCH(24);
  else BP_LAUNCH(32);
#undef BP_LAUNCH
  return hipGetLastError();
}

}  // namespace empose
