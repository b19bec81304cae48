// The per-frame bodies of the pose / shape update + Rodrigues + feature row (update_feat_kernel, smpl.hip) and of the
// Rodrigues reverse in tile layout (rodrigues_bwd_t_kernel, smpl_tile.hip), shared with the blend-shape GEMMs that
// carry them as prologue / epilogue (mlp_fused.hip: blend_feat_gemm_kernel, blend_t_gemm_rod_kernel).  One definition,
// so the stand-alone kernels and the fused ones produce the same bits.  Internal, gfx950 only.
#pragma once
#include "kernels.h"
#include "smpl_math.h"

namespace empose {

// 32 lanes per frame: slots 0..21 are joints, 22..31 the ten shape coefficients.
//
// Window mean of the shape update (shape_avg) for the window that starts at frame w0, by 32 lanes (all of them must
// call): lane `slot` fetches the frames slot, slot + 32, ... of the window -- independent loads instead of a serial walk
// by the ten shape lanes -- and the sums meet by shuffles.  Returns, on lane NB + k, the mean of coefficient k.
//   shape_avg == 1: mean over ALL frames of the window incl. padded ones (reference models.py:529-532);
//   shape_avg == 2: mean over the valid frames only (what an unpadded window of that length would give; used by the
//   batched streaming driver so that ragged batches reproduce one-recording-at-a-time results)
__device__ __forceinline__ float shape_window_mean(const FeatArgs& a, int window, int slot) {
  const int w0 = window * a.F;
  const int n = (a.shape_avg == 2 && a.seq_lengths) ? max(1, min(a.F, a.seq_lengths[window])) : a.F;
  float part[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) part[k] = 0.f;
  for (int f = slot; f < n; f += 32) {   // ten independent loads per trip (long windows: F = 256 is eight trips)
    const float* row = a.d_beta + (size_t)(w0 + f) * 10;
#pragma unroll
    for (int k = 0; k < 10; ++k) part[k] += row[k];
  }
  float d_mean = 0.f;
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    float v = part[k];
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) v += __shfl_xor(v, off, 32);
    if (slot == NB + k) d_mean = v / (float)n;
  }
  return d_mean;
}

// What a lane brings to its frame: the updated axis-angle of joint `slot` (lanes < NB) or the updated shape
// coefficient slot - NB.  Loads and arithmetic only (no store), so that a caller with several frames per lane can have
// all of them in flight before the first is used.
struct FeatLane { float r0, r1, r2; };
__device__ __forceinline__ FeatLane feat_lane_load(const FeatArgs& a, int t, int slot, float d_mean) {
  FeatLane v{0.f, 0.f, 0.f};
  if (t >= a.T) return v;
  if (slot < NB) {
    const float* th = a.theta + (size_t)t * a.ld_theta + slot * 3;
    v.r0 = th[0]; v.r1 = th[1]; v.r2 = th[2];
    if (a.d_theta) {
      const float* d = a.d_theta + (size_t)t * 66 + slot * 3;
      v.r0 = v.r0 + d[0] * a.theta_step;
      v.r1 = v.r1 + d[1] * a.theta_step;
      v.r2 = v.r2 + d[2] * a.theta_step;
    }
  } else {
    const int k = slot - NB;
    float b = a.beta_keep != 0.f ? a.beta[(size_t)t * a.ld_beta + k] * a.beta_keep : 0.f;
    if (a.d_beta) {
      const float d = a.shape_avg ? d_mean : a.d_beta[(size_t)t * 10 + k];
      b = b + d * a.beta_step;
    }
    v.r0 = b;
  }
  return v;
}
// The stores of frame t (the caller's rows and the optional copies), Rodrigues and the 200-float feature row
// [vec(R_j - I), j = 1..21 | beta | 1] to feat_row; when rot_row is given, the 22 rotations (198 floats) too.  Frames
// past the end (t >= a.T) leave everything untouched.
// FAST: Rodrigues with the short sine / cosine of smpl_math.h (what the frame-per-lane kernel uses for the chain's
// rotations of the same frames; ~1 ulp from the library's) -- the GEMM prologue has nothing to hide ~150 dependent
// instructions behind.
template <bool FAST = false>
__device__ __forceinline__ void feat_lane_finish(const FeatArgs& a, int t, int slot, const FeatLane& v, float* feat_row,
                                                 float* rot_row) {
  if (t >= a.T) return;
  if (slot < NB) {
    const float r0 = v.r0, r1 = v.r1, r2 = v.r2;
    if (a.d_theta) {
      float* th = a.theta + (size_t)t * a.ld_theta + slot * 3;
      th[0] = r0; th[1] = r1; th[2] = r2;
    }
    if (a.out_theta) { float* o = a.out_theta + (size_t)t * 66 + slot * 3; o[0] = r0; o[1] = r1; o[2] = r2; }
    if (a.out_theta2) { float* o = a.out_theta2 + (size_t)t * 66 + slot * 3; o[0] = r0; o[1] = r1; o[2] = r2; }
    if (a.theta_t) {
      float* o = a.theta_t + ((size_t)(t >> 6) * 66 + slot * 3) * 64 + (t & 63);
      o[0] = r0; o[64] = r1; o[128] = r2;
    }
    Rod q; float R[9];
    if (FAST) rodrigues_fast(r0, r1, r2, a.rod_conv, q, R);
    else rodrigues(r0, r1, r2, a.rod_conv, q, R);
    if (rot_row) {
      float* ro = rot_row + slot * 9;
#pragma unroll
      for (int e = 0; e < 9; ++e) ro[e] = R[e];
    }
    if (slot >= 1) {
      float* f = feat_row + (slot - 1) * 9;
      f[0] = R[0] - 1.f; f[1] = R[1]; f[2] = R[2];
      f[3] = R[3]; f[4] = R[4] - 1.f; f[5] = R[5];
      f[6] = R[6]; f[7] = R[7]; f[8] = R[8] - 1.f;
    }
  } else {
    const int k = slot - NB;
    const float b = v.r0;
    // The lanes of a window read d_beta of all its frames but write only beta[t][k]: no hazard.
    if (a.d_beta || a.beta_keep != 1.f) a.beta[(size_t)t * a.ld_beta + k] = b;   // plain evaluation: the caller's rows are left alone
    if (a.out_beta) a.out_beta[(size_t)t * 10 + k] = b;
    if (a.out_beta2) a.out_beta2[(size_t)t * 10 + k] = b;
    feat_row[189 + k] = b;
    if (k == 0) feat_row[199] = 1.f;
  }
}
// One frame by its 32 lanes (update_feat_kernel): all 32 lanes must call it.
__device__ __forceinline__ void feat_frame(const FeatArgs& a, int t, int slot, float* feat_row, float* rot_row) {
  float d_mean = 0.f;
  if (a.d_beta && a.shape_avg) d_mean = shape_window_mean(a, (t < a.T ? t : a.T - 1) / a.F, slot);
  const FeatLane v = feat_lane_load(a, t, slot, d_mean);
  feat_lane_finish(a, t, slot, v, feat_row, rot_row);
}

// d R_j -> d theta_j for one (frame, joint): dR = the chain's cotangent + the feature cotangent (joints >= 1).
template <bool FAST = false>
__device__ __forceinline__ void rodrigues_bwd_joint(const float (&th)[3], int rod_conv, const float (&dR)[9], float (&g)[3]) {
  Rod q; float R[9];
  if (FAST) rodrigues_fast(th[0], th[1], th[2], rod_conv, q, R);
  else rodrigues(th[0], th[1], th[2], rod_conv, q, R);
  const float K[9] = {0.f, -q.dz, q.dy, q.dz, 0.f, -q.dx, -q.dy, q.dx, 0.f};
  const float KK[9] = {-q.dz * q.dz - q.dy * q.dy, q.dx * q.dy, q.dx * q.dz,
                       q.dx * q.dy, -q.dz * q.dz - q.dx * q.dx, q.dy * q.dz,
                       q.dx * q.dz, q.dy * q.dz, -q.dy * q.dy - q.dx * q.dx};
  float ds = 0.f, dc1 = 0.f;
#pragma unroll
  for (int e = 0; e < 9; ++e) { ds += dR[e] * K[e]; dc1 += dR[e] * KK[e]; }
  const float oc = 1.f - q.c;
  float dK[9];   // dK = s dR + (1-c) (dR K^T + K^T dR)
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float mm = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) mm += dR[r * 3 + k] * K[c * 3 + k] + K[k * 3 + r] * dR[k * 3 + c];
      dK[r * 3 + c] = q.s * dR[r * 3 + c] + oc * mm;
    }
  const float ddx = dK[7] - dK[5], ddy = dK[2] - dK[6], ddz = dK[3] - dK[1];
  float da = ds * q.c + dc1 * q.s;
  da -= (ddx * q.dx + ddy * q.dy + ddz * q.dz) / q.ang;
  g[0] = ddx / q.ang + da * q.ux / q.ang;
  g[1] = ddy / q.ang + da * q.uy / q.ang;
  g[2] = ddz / q.ang + da * q.uz / q.ang;
}

// The Rodrigues reverse of a 64-frame tile by a 256-thread workgroup, lane = frame, the waves share the joints:
// d_rot in tile layout from global memory, the feature cotangents through `df(column)` (global tile layout or LDS), the 76
// outputs of a frame through `sg` (LDS, 64 x 77 floats) as contiguous row pieces (the caller's rows have a stride of
// ~300 floats).  Ends with the rows written; contains one __syncthreads.
template <bool FAST = false, class DF>
__device__ __forceinline__ void rodrigues_bwd_tile(const RodBwdTArgs& a, int tile, float* sg, DF df) {
  constexpr int FR = TL_FR, LD = ROD_BWD_LD;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int t = tile * FR + lane;
  const int tc = t < a.T ? t : a.T - 1;
  const float* dr_t = a.d_rot_t + (size_t)tile * (NB * 9) * FR + lane;
  const float* th_t = a.theta_t ? a.theta_t + (size_t)tile * 66 * FR + lane : nullptr;   // coalesced when the caller has it
  // the joints of this wave: wave, wave + 4, ... ; all their loads first, then the arithmetic
  constexpr int NJ = (NB + 3) / 4;
  float dR[NJ][9], th[NJ][3];
#pragma unroll
  for (int u = 0; u < NJ; ++u) {
    const int j = wave + 4 * u;
    const int jc = j < NB ? j : NB - 1;
#pragma unroll
    for (int e = 0; e < 9; ++e) dR[u][e] = dr_t[(size_t)(jc * 9 + e) * FR];
#pragma unroll
    for (int e = 0; e < 3; ++e)
      th[u][e] = th_t ? th_t[(size_t)(jc * 3 + e) * FR] : a.theta[(size_t)tc * a.ld_theta + jc * 3 + e];
  }
#pragma unroll
  for (int u = 0; u < NJ; ++u) {
    const int j = wave + 4 * u;
    if (j >= NB) break;
    if (j >= 1) {
#pragma unroll
      for (int e = 0; e < 9; ++e) dR[u][e] += df((j - 1) * 9 + e);
    }
    float g[3];
    rodrigues_bwd_joint<FAST>(th[u], a.rod_conv, dR[u], g);
    sg[lane * LD + j * 3 + 0] = g[0];
    sg[lane * LD + j * 3 + 1] = g[1];
    sg[lane * LD + j * 3 + 2] = g[2];
  }
  for (int k = wave; k < 10; k += 4) sg[lane * LD + 66 + k] = df(189 + k);
  __syncthreads();
  // rows out: thread -> (frame, column), consecutive threads consecutive columns of one frame
  for (int i = threadIdx.x; i < FR * 76; i += 256) {
    const int f = i / 76, c = i - f * 76;
    const int tt = tile * FR + f;
    if (tt >= a.T) continue;
    const float val = sg[f * LD + c];
    if (c < 66) {
      a.g_theta[(size_t)tt * a.ld_g + c] = val;
      if (a.trace_g_theta) a.trace_g_theta[(size_t)tt * 66 + c] = val;
    } else {
      a.g_beta[(size_t)tt * a.ld_gb + (c - 66)] = val;
      if (a.trace_g_beta) a.trace_g_beta[(size_t)tt * 10 + (c - 66)] = val;
    }
  }
}

}  // namespace empose
