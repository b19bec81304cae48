// Train-mode MLP layer, fused (round 3; reference nn/layers.py:13-77 in training mode: Linear -> BatchNorm1d -> PReLU).
//
// Layer by layer the training step ran  GEMM -> HBM -> column statistics -> combine -> transform -> HBM  forward and
// GEMM -> HBM -> statistics -> combine -> transform -> HBM  backward: six memory-bound passes per layer application
// (1.84 ms of the 11.3 ms step at 256 windows, profiles/r02f_train_kernel_stats_bs256.csv).  Here the passes ride on
// the GEMMs that produce / consume the data:
//
//   forward   y_l = a_{l-1} W_l^T + b_l           the GEMM's epilogue also emits, per 32-row block and column, the sum
//                                                 and the centred sum of squares of y (combined in block order with
//                                                 Chan's update: deterministic, no cancellation);
//             a_l = PReLU(s_l y_l + t_l)          never materialised: s = gamma rstd, t = beta - mean s are applied
//                                                 while the NEXT GEMM stages its A operand (and wherever else a_l is
//                                                 read: the weight-gradient product).  Saved per layer: y_l only.
//   backward  dA_l = dY_{l+1} W_{l+1}             the GEMM's epilogue turns it into dyh = dA * PReLU'(yhat) and emits the
//                                                 column sums of dyh, dyh * xhat and (yhat <= 0) dA * yhat (-> dbeta,
//                                                 dgamma, dslope);
//             dY_l = c1 dyh + c3 y_l + c0         (BatchNorm reverse, per-column coefficients from those sums) in ONE pass,
//                                                 in place.  (Forming dY inside the operand staging of its two
//                                                 consumers was measured too: both then read y a second time, the
//                                                 A^T B product ran 209 instead of 163 us -- more than the pass costs.)
//
// The GEMM side lives in gemm_f32.hip (gemm_tn_f32_kernel<CfgS12, 2>: the training step's tuned 64 x 128 tile with an
// optional A-operand transform and the two train epilogues; a separate, simpler tile kernel was 27 % slower than the
// tuned one and ate the gain).  This file: the small kernels between the GEMMs.
#include "kernels.h"

#include "gemm_epilogue.h"

#include <algorithm>

namespace empose {

typedef float f4v __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------------
// Combine kernels: a workgroup per 64 columns; sixteen groups each walk a contiguous share of the row blocks in order,
// the groups are then merged in order -- a fixed tree, so the results are reproducible.
// ---------------------------------------------------------------------------------------------------------------
constexpr int BNF_G = 16;
// forward: (sum, centred sum of squares) per 32-row block -> mean, rstd (Chan's parallel update: no cancellation), the
// fused transform coefficients s = gamma rstd, t = beta - mean s, running statistics (momentum, unbiased variance, as
// torch.nn.BatchNorm1d)
__global__ __launch_bounds__(64 * BNF_G) void bn_fused_combine_fwd_kernel(BnFusedFwdArgs a) {
  __shared__ float red[3][BNF_G][64];
  const int lc = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lc;
  const int cc = c < a.C ? c : a.C - 1;
  const int nb = (a.M + 31) / 32, per = (nb + BNF_G - 1) / BNF_G;
  const int b0 = g * per, b1 = min(nb, b0 + per);
  float n = 0.f, mean = 0.f, m2 = 0.f;
  // full blocks have 32 rows: sum of the block sums and of the centred sums, then the between-block term from the
  // block means (exactly Chan's merge for equal counts: M2 = sum M2_b + 32 sum (mean_b - mean_g)^2); a last partial
  // block is merged in afterwards
  float s1 = 0.f, s2 = 0.f, sq = 0.f;
  int nfull = 0;
  for (int b = b0; b < b1; ++b) {
    if (a.M - 32 * b < 32) break;
    const float* p = a.part + (size_t)b * 2 * a.C;
    const float mb = p[cc] * (1.f / 32.f);
    s1 += mb; s2 += p[a.C + cc]; sq += mb * mb;
    ++nfull;
  }
  if (nfull > 0) {
    n = 32.f * (float)nfull;
    mean = s1 / (float)nfull;
    m2 = s2 + 32.f * fmaxf(sq - (float)nfull * mean * mean, 0.f);
  }
  if (b0 + nfull < b1) {   // the matrix's last, partial block
    const int b = b0 + nfull;
    const float nbk = (float)(a.M - 32 * b);
    const float* p = a.part + (size_t)b * 2 * a.C;
    const float mb = p[cc] / nbk, delta = mb - mean, tot = n + nbk;
    mean += delta * (nbk / tot);
    m2 += p[a.C + cc] + delta * delta * (n * nbk / tot);
    n = tot;
  }
  red[0][g][lc] = n; red[1][g][lc] = mean; red[2][g][lc] = m2;
  __syncthreads();
  if (g != 0 || c >= a.C) return;
  n = 0.f; mean = 0.f; m2 = 0.f;
#pragma unroll
  for (int k = 0; k < BNF_G; ++k) {
    const float nk = red[0][k][lc];
    if (nk > 0.f) {
      const float delta = red[1][k][lc] - mean, tot = n + nk;
      mean += delta * (nk / tot);
      m2 += red[2][k][lc] + delta * delta * (n * nk / tot);
      n = tot;
    }
  }
  const float var = m2 / (float)a.M;   // biased: what normalises the batch
  const float rstd = 1.f / sqrtf(var + a.eps);
  const float s = a.gamma[c] * rstd;
  a.mean[c] = mean; a.rstd[c] = rstd; a.s[c] = s; a.t[c] = a.beta[c] - mean * s;
  if (a.running_mean) {
    const float unbiased = a.M > 1 ? var * (float)a.M / (float)(a.M - 1) : var;
    a.running_mean[c] = (1.f - a.momentum) * a.running_mean[c] + a.momentum * mean;
    a.running_var[c] = (1.f - a.momentum) * a.running_var[c] + a.momentum * unbiased;
  }
  if (c == 0 && a.num_batches_tracked) a.num_batches_tracked[0] += 1;
}

// backward: block sums -> dbeta, dgamma (+ accumulate), the workgroup's share of the slope gradient, and the
// coefficients of  dY = c1 dyh + c3 y + c0  (BatchNorm reverse:  gamma rstd / M (M dyh - dbeta - xhat dgamma))
__global__ __launch_bounds__(64 * BNF_G) void bn_fused_combine_bwd_kernel(BnFusedBwdArgs a) {
  __shared__ float red[3][BNF_G][64];
  const int lc = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lc;
  const bool ok = c < a.C;
  const int cc = ok ? c : a.C - 1;
  const int nb = (a.M + 31) / 32, per = (nb + BNF_G - 1) / BNF_G;
  const int b0 = g * per, b1 = min(nb, b0 + per);
  float sb = 0.f, sg = 0.f, sa = 0.f;
  for (int b = b0; b < b1; ++b) {
    const float* p = a.part + (size_t)b * 3 * a.C;
    sb += p[cc]; sg += p[a.C + cc]; sa += p[2 * a.C + cc];
  }
  red[0][g][lc] = sb; red[1][g][lc] = sg; red[2][g][lc] = ok ? sa : 0.f;
  __syncthreads();
  if (g != 0) return;
  sb = 0.f; sg = 0.f; sa = 0.f;
#pragma unroll
  for (int k = 0; k < BNF_G; ++k) { sb += red[0][k][lc]; sg += red[1][k][lc]; sa += red[2][k][lc]; }
  if (ok) {
    a.dgamma[c] = sg + (a.accumulate ? a.dgamma[c] : 0.f);
    a.dbeta[c] = sb + (a.accumulate ? a.dbeta[c] : 0.f);
    const float gm = a.gamma[c], rstd = a.rstd[c], mean = a.mean[c], inv_m = 1.f / (float)a.M;
    const float c1 = gm * rstd, c3 = -gm * rstd * rstd * sg * inv_m;
    a.coef[c] = c1;
    a.coef[a.C + c] = c3;
    a.coef[2 * a.C + c] = -c1 * sb * inv_m - c3 * mean;
  }
  float t = sa;   // slope: this workgroup's 64 columns in lane order (one wave)
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) t += __shfl_xor(t, off, 64);
  if (lc == 0) a.dslope_partial[blockIdx.x] = t;
}
// the slope gradient: the combine workgroups' shares in order
__global__ void bn_fused_slope_kernel(const float* partial, int n, float* dslope, int accumulate) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  float t = 0.f;
  for (int i = 0; i < n; ++i) t += partial[i];
  dslope[0] = t + (accumulate ? dslope[0] : 0.f);
}

// dY = c1 dyh + c3 y + c0, in place over dyh (16-byte pieces; C % 4 == 0).  Measured: forming dY inside the operand
// staging of its two consumers (the dX GEMM and the A^T B product each read y a second time) cost more than this pass.
__global__ __launch_bounds__(256) void bn_fused_apply_bwd_kernel(float* __restrict__ dyh, const float* __restrict__ y,
                                                                 const float* __restrict__ coef, int M, int C) {
  const int c4n = C >> 2;
  const size_t n = (size_t)M * c4n;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % c4n) * 4;
    const f4v c1 = *reinterpret_cast<const f4v*>(coef + c), c3 = *reinterpret_cast<const f4v*>(coef + C + c),
              c0 = *reinterpret_cast<const f4v*>(coef + 2 * C + c);
    f4v d = reinterpret_cast<f4v*>(dyh)[i];
    const f4v yy = reinterpret_cast<const f4v*>(y)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) d[e] = c1[e] * d[e] + (c3[e] * yy[e] + c0[e]);
    reinterpret_cast<f4v*>(dyh)[i] = d;
  }
}
hipError_t launch_bn_fused_apply_bwd(float* dyh, const float* y, const float* coef, int M, int C, hipStream_t stream) {
  const size_t n = (size_t)M * (C >> 2);
  const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(bn_fused_apply_bwd_kernel, dim3(blocks), dim3(256), 0, stream, dyh, y, coef, M, C);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Round 4: the "finish" kernels -- statistics from the GEMM epilogues as above, but the layer's activation / cotangent IS
// materialised (the next GEMM and the weight-gradient product read a ready operand; forming it inside their operand
// staging costs the transform once per column-tile workgroup, measured slower), and combine + apply are ONE launch:
//   forward   y (GEMM, column statistics per 32-row block in its epilogue)  ->  a = PReLU(s y + t)
//   backward  dyh (GEMM epilogue: dA * PReLU', three column sums per block) ->  dY = c1 dyh + c3 y + c0 in place
// A workgroup owns a stripe of 32 columns x a block of rows.  It first combines the per-block partial sums of ITS 32
// columns itself (8 groups walk the blocks in order, merged in order: every workgroup of a stripe computes bit-identical
// coefficients; the partials are L2-resident, 8 KB per 32 blocks) and then streams its rows.  The workgroups of row
// block 0 also write the statistics / parameter gradients of their stripe.  Two launches per layer and direction
// instead of four (GEMM, statistics, combine, apply).
// ---------------------------------------------------------------------------------------------------------------
constexpr int BNX_W = 32, BNX_G = 8;   // columns per workgroup, groups of partial-sum blocks

__global__ __launch_bounds__(256) void bn_finish_fwd_kernel(BnFinishFwdArgs a) {
  __shared__ float red[3][BNX_G][BNX_W];
  __shared__ __attribute__((aligned(16))) float st[2][BNX_W];
  const int lc = threadIdx.x & (BNX_W - 1), g = threadIdx.x / BNX_W;
  const int c = blockIdx.x * BNX_W + lc;
  const int cc = c < a.C ? c : a.C - 1;
  {
    const int nb = (a.M + 31) / 32, per = (nb + BNX_G - 1) / BNX_G;
    const int b0 = g * per, b1 = min(nb, b0 + per);
    float n = 0.f, mean = 0.f, m2 = 0.f, s1 = 0.f, s2 = 0.f, sq = 0.f;
    // (see bn_fused_combine_fwd_kernel: Chan's merge for equal counts, then the matrix's last, partial block.)  The full
    // blocks of this group are a plain counted loop -- its loads are independent, eight blocks in flight: the combine
    // is latency, not bandwidth (16 KB per workgroup), and every workgroup waits for it before it streams its rows
    const int nfull = max(0, min(b1, a.M / 32) - b0);
    const float* p0 = a.part + (size_t)b0 * 2 * a.C + cc;
    const size_t bs = (size_t)2 * a.C;
    int b = 0;
    for (; b + 8 <= nfull; b += 8) {
      float v1[8], v2[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { v1[u] = p0[(b + u) * bs]; v2[u] = p0[(b + u) * bs + a.C]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float mb = v1[u] * (1.f / 32.f);
        s1 += mb; s2 += v2[u]; sq += mb * mb;
      }
    }
    for (; b < nfull; ++b) {
      const float mb = p0[b * bs] * (1.f / 32.f);
      s1 += mb; s2 += p0[b * bs + a.C]; sq += mb * mb;
    }
    if (nfull > 0) {
      n = 32.f * (float)nfull;
      mean = s1 / (float)nfull;
      m2 = s2 + 32.f * fmaxf(sq - (float)nfull * mean * mean, 0.f);
    }
    if (b0 + nfull < b1) {
      const int bl = b0 + nfull;
      const float nbk = (float)(a.M - 32 * bl);
      const float* p = a.part + (size_t)bl * 2 * a.C;
      const float mb = p[cc] / nbk, delta = mb - mean, tot = n + nbk;
      mean += delta * (nbk / tot);
      m2 += p[a.C + cc] + delta * delta * (n * nbk / tot);
      n = tot;
    }
    red[0][g][lc] = n; red[1][g][lc] = mean; red[2][g][lc] = m2;
  }
  __syncthreads();
  if (g == 0) {
    float n = 0.f, mean = 0.f, m2 = 0.f;
#pragma unroll
    for (int k = 0; k < BNX_G; ++k) {
      const float nk = red[0][k][lc];
      if (nk > 0.f) {
        const float delta = red[1][k][lc] - mean, tot = n + nk;
        mean += delta * (nk / tot);
        m2 += red[2][k][lc] + delta * delta * (n * nk / tot);
        n = tot;
      }
    }
    const float var = m2 / (float)a.M;   // biased: what normalises the batch
    const float rstd = 1.f / sqrtf(var + a.eps);
    const float s = a.gamma[cc] * rstd, t = a.beta[cc] - mean * s;
    st[0][lc] = s; st[1][lc] = t;
    if (blockIdx.y == 0 && c < a.C) {
      a.mean[c] = mean; a.rstd[c] = rstd; a.s[c] = s; a.t[c] = t;
      if (a.running_mean) {
        const float unbiased = a.M > 1 ? var * (float)a.M / (float)(a.M - 1) : var;
        a.running_mean[c] = (1.f - a.momentum) * a.running_mean[c] + a.momentum * mean;
        a.running_var[c] = (1.f - a.momentum) * a.running_var[c] + a.momentum * unbiased;
      }
      if (c == 0 && a.num_batches_tracked) a.num_batches_tracked[0] += 1;
    }
  }
  __syncthreads();
  // ---- apply: 8 lanes x 16 bytes cover the stripe's 32 columns of a row, 32 rows per pass
  const int q = threadIdx.x & 7, r0 = threadIdx.x >> 3;
  const int c4 = blockIdx.x * BNX_W + q * 4;
  if (c4 >= a.C) return;   // (C % 4 == 0)
  const f4v s4 = *reinterpret_cast<const f4v*>(&st[0][q * 4]), t4 = *reinterpret_cast<const f4v*>(&st[1][q * 4]);
  const float slope = a.slope[0];
  const int m_end = min(a.M, (int)(blockIdx.y + 1) * a.rows_per_block);
  for (int m = blockIdx.y * a.rows_per_block + r0; m < m_end; m += 32) {
    const f4v y = *reinterpret_cast<const f4v*>(a.y + (size_t)m * a.ldy + c4);
    f4v o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float v = s4[e] * y[e] + t4[e];
      o[e] = v > 0.f ? v : slope * v;
    }
    *reinterpret_cast<f4v*>(a.act + (size_t)m * a.ld_act + c4) = o;
  }
}

__global__ __launch_bounds__(256) void bn_finish_bwd_kernel(BnFinishBwdArgs a) {
  __shared__ float red[3][BNX_G][BNX_W];
  __shared__ __attribute__((aligned(16))) float cf[3][BNX_W];
  const int lc = threadIdx.x & (BNX_W - 1), g = threadIdx.x / BNX_W;
  const int c = blockIdx.x * BNX_W + lc;
  const bool ok = c < a.C;
  const int cc = ok ? c : a.C - 1;
  {
    const int nb = (a.M + 31) / 32, per = (nb + BNX_G - 1) / BNX_G;
    const int b0 = g * per, b1 = min(nb, b0 + per);
    float sb = 0.f, sg = 0.f, sa = 0.f;
    const float* p0 = a.part + (size_t)b0 * 3 * a.C + cc;
    const size_t bs = (size_t)3 * a.C;
    const int cnt = max(0, b1 - b0);
    int b = 0;
    for (; b + 8 <= cnt; b += 8) {   // eight blocks in flight (the combine is latency; see bn_finish_fwd_kernel)
      float v0[8], v1[8], v2[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { v0[u] = p0[(b + u) * bs]; v1[u] = p0[(b + u) * bs + a.C]; v2[u] = p0[(b + u) * bs + 2 * a.C]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) { sb += v0[u]; sg += v1[u]; sa += v2[u]; }
    }
    for (; b < cnt; ++b) { sb += p0[b * bs]; sg += p0[b * bs + a.C]; sa += p0[b * bs + 2 * a.C]; }
    red[0][g][lc] = sb; red[1][g][lc] = sg; red[2][g][lc] = ok ? sa : 0.f;
  }
  __syncthreads();
  if (g == 0) {
    float sb = 0.f, sg = 0.f, sa = 0.f;
#pragma unroll
    for (int k = 0; k < BNX_G; ++k) { sb += red[0][k][lc]; sg += red[1][k][lc]; sa += red[2][k][lc]; }
    const float gm = a.gamma[cc], rstd = a.rstd[cc], mean = a.mean[cc], inv_m = 1.f / (float)a.M;
    const float c1 = gm * rstd, c3 = -gm * rstd * rstd * sg * inv_m;
    cf[0][lc] = c1; cf[1][lc] = c3; cf[2][lc] = -c1 * sb * inv_m - c3 * mean;
    if (blockIdx.y == 0) {
      if (ok) {
        a.dgamma[c] = sg + (a.accumulate ? a.dgamma[c] : 0.f);
        a.dbeta[c] = sb + (a.accumulate ? a.dbeta[c] : 0.f);
      }
      // the slope gradient: this stripe's 32 columns in lane order, then the stripes in index order by whichever stripe
      // arrives last (deterministic; the counter re-arms itself)
      float t = sa;
#pragma unroll
      for (int off = 16; off >= 1; off >>= 1) t += __shfl_xor(t, off, 32);
      if (lc == 0) {
        __hip_atomic_store(a.dslope_partial + blockIdx.x, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        if (atomicAdd(a.counter, 1) == (int)gridDim.x - 1) {
          __threadfence();
          float total = 0.f;
          for (unsigned i = 0; i < gridDim.x; ++i)
            total += __hip_atomic_load(a.dslope_partial + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          a.dslope[0] = total + (a.accumulate ? a.dslope[0] : 0.f);
          a.counter[0] = 0;
        }
      }
    }
  }
  __syncthreads();
  const int q = threadIdx.x & 7, r0 = threadIdx.x >> 3;
  const int c4 = blockIdx.x * BNX_W + q * 4;
  if (c4 >= a.C) return;
  const f4v c1 = *reinterpret_cast<const f4v*>(&cf[0][q * 4]), c3 = *reinterpret_cast<const f4v*>(&cf[1][q * 4]),
            c0 = *reinterpret_cast<const f4v*>(&cf[2][q * 4]);
  const int m_end = min(a.M, (int)(blockIdx.y + 1) * a.rows_per_block);
  for (int m = blockIdx.y * a.rows_per_block + r0; m < m_end; m += 32) {
    f4v d = *reinterpret_cast<const f4v*>(a.dyh + (size_t)m * a.ld + c4);
    const f4v yy = *reinterpret_cast<const f4v*>(a.y + (size_t)m * a.ldy + c4);
#pragma unroll
    for (int e = 0; e < 4; ++e) d[e] = c1[e] * d[e] + (c3[e] * yy[e] + c0[e]);
    *reinterpret_cast<f4v*>(a.dyh + (size_t)m * a.ld + c4) = d;
  }
}

// rows per workgroup: enough workgroups to fill the CUs (stripes x row blocks >= ~512), at least 64 rows each
static int bn_finish_rows_per_block(int M, int C) {
  const int stripes = (C + BNX_W - 1) / BNX_W;
  int blocks_y = std::max(1, 512 / stripes);
  blocks_y = std::min(blocks_y, std::max(1, M / 64));
  const int rows = (M + blocks_y - 1) / blocks_y;
  return (rows + 31) & ~31;
}
hipError_t launch_bn_finish_fwd(BnFinishFwdArgs a, hipStream_t stream) {
  if (a.C % 4 != 0 || a.ldy % 4 != 0 || a.ld_act % 4 != 0) return hipErrorInvalidValue;
  a.rows_per_block = bn_finish_rows_per_block(a.M, a.C);
  const dim3 grid((a.C + BNX_W - 1) / BNX_W, (a.M + a.rows_per_block - 1) / a.rows_per_block);
  hipLaunchKernelGGL(bn_finish_fwd_kernel, grid, dim3(256), 0, stream, a);
  return hipGetLastError();
}
hipError_t launch_bn_finish_bwd(BnFinishBwdArgs a, hipStream_t stream) {
  if (a.C % 4 != 0 || a.ldy % 4 != 0 || a.ld % 4 != 0) return hipErrorInvalidValue;
  a.rows_per_block = bn_finish_rows_per_block(a.M, a.C);
  const dim3 grid((a.C + BNX_W - 1) / BNX_W, (a.M + a.rows_per_block - 1) / a.rows_per_block);
  hipLaunchKernelGGL(bn_finish_bwd_kernel, grid, dim3(256), 0, stream, a);
  return hipGetLastError();
}

size_t bn_fused_partial_floats(int M, int C) { return (size_t)((M + 31) / 32) * 3 * C; }

hipError_t launch_bn_fused_combine_fwd(const BnFusedFwdArgs& a, hipStream_t stream) {
  hipLaunchKernelGGL(bn_fused_combine_fwd_kernel, dim3((a.C + 63) / 64), dim3(64 * BNF_G), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_bn_fused_combine_bwd(const BnFusedBwdArgs& a, hipStream_t stream) {
  const int nwg = (a.C + 63) / 64;
  hipLaunchKernelGGL(bn_fused_combine_bwd_kernel, dim3(nwg), dim3(64 * BNF_G), 0, stream, a);
  hipLaunchKernelGGL(bn_fused_slope_kernel, dim3(1), dim3(1), 0, stream, a.dslope_partial, nwg, a.dslope, a.accumulate);
  return hipGetLastError();
}

}  // namespace empose
