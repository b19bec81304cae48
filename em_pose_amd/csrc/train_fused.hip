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
