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

__global__ __launch_bounds__(bp::NT) void bn_prelu_fwd_kernel(BnPreluArgs a) {
  using namespace bp;
  __shared__ float red[RG * COLS];
  const int c = threadIdx.x & (COLS - 1), rg = threadIdx.x / COLS;
  const int col = blockIdx.x * COLS + c;
  const bool ok = col < a.C;
  const int cc = ok ? col : a.C - 1;
  const float inv_m = 1.f / (float)a.M;
  float s = 0.f;
#pragma unroll 4
  for (int m = rg; m < a.M; m += RG) s += a.x[(size_t)m * a.ldx + cc];
  const float mean = bp_reduce_rows(s, red, rg, c) * inv_m;
  float q = 0.f;
#pragma unroll 4
  for (int m = rg; m < a.M; m += RG) { const float d = a.x[(size_t)m * a.ldx + cc] - mean; q += d * d; }
  const float var = bp_reduce_rows(q, red, rg, c) * inv_m;      // biased: what normalises the batch
  const float rstd = 1.f / sqrtf(var + a.eps);
  const float g = a.gamma[cc], b = a.beta[cc], slope = a.slope[0];
  if (ok) {
#pragma unroll 4
    for (int m = rg; m < a.M; m += RG) {
      const float y = g * ((a.x[(size_t)m * a.ldx + col] - mean) * rstd) + b;
      a.z[(size_t)m * a.ldz + col] = y > 0.f ? y : slope * y;
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

__global__ __launch_bounds__(bp::NT) void bn_prelu_bwd_kernel(BnPreluArgs a) {
  using namespace bp;
  __shared__ float red[RG * COLS];
  const int c = threadIdx.x & (COLS - 1), rg = threadIdx.x / COLS;
  const int col = blockIdx.x * COLS + c;
  const bool ok = col < a.C;
  const int cc = ok ? col : a.C - 1;
  const float mean = a.save_mean[cc], rstd = a.save_rstd[cc];
  const float g = a.gamma[cc], b = a.beta[cc], slope = a.slope[0];
  float s_b = 0.f, s_g = 0.f, s_a = 0.f;
#pragma unroll 4
  for (int m = rg; m < a.M; m += RG) {
    const float xh = (a.x[(size_t)m * a.ldx + cc] - mean) * rstd;
    const float y = g * xh + b;
    const float dz = a.dz[(size_t)m * a.lddz + cc];
    const float dy = y > 0.f ? dz : slope * dz;
    s_b += dy;
    s_g += dy * xh;
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
      a.dslope[0] = total;
      a.counter[0] = 0;
    }
  }
  if (!ok) return;
  const float k = g * rstd / (float)a.M;
#pragma unroll 4
  for (int m = rg; m < a.M; m += RG) {
    const float xh = (a.x[(size_t)m * a.ldx + col] - mean) * rstd;
    const float y = g * xh + b;
    const float dz = a.dz[(size_t)m * a.lddz + col];
    const float dy = y > 0.f ? dz : slope * dz;
    a.dx[(size_t)m * a.lddx + col] = k * ((float)a.M * dy - dbeta - xh * dgamma);
  }
  if (rg == 0) { a.dgamma[col] = dgamma; a.dbeta[col] = dbeta; }
}

hipError_t launch_bn_prelu(const BnPreluArgs& a, bool backward, hipStream_t stream) {
  const int blocks = (a.C + bp::COLS - 1) / bp::COLS;
  if (backward) hipLaunchKernelGGL(bn_prelu_bwd_kernel, dim3(blocks), dim3(bp::NT), 0, stream, a);
  else hipLaunchKernelGGL(bn_prelu_fwd_kernel, dim3(blocks), dim3(bp::NT), 0, stream, a);
  return hipGetLastError();
}

}  // namespace empose
