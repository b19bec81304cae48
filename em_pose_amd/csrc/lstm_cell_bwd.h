// One element (batch row b, hidden unit j) of a step of back-propagation through time of an LSTM layer (see the comment of
// lstm_cell_bwd_kernel in train.hip).  Shared by that kernel and by the reduce kernel of the K-split recurrent product
// (gemm_f32.hip), which feeds the dh it has just formed straight into the next step's cell instead of storing it.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace empose {

// Everything the element reads that does not depend on the incoming dh: fetched BEFORE the recurrent product that forms dh
// (rec_fewrows_kernel), so that the step's tail is arithmetic and stores only.
struct LstmCellBwdPre {
  float dc_in, gi, gf, gg, go, tc, c_prev, dy;
  int live;
};
__device__ __forceinline__ LstmCellBwdPre lstm_cell_bwd_load(const LstmCellBwdArgs& a, int idx) {
  const int b = idx / a.H, j = idx - b * a.H;
  const int H = a.H, F = a.F, t = a.t;
  const int len = a.seq_lengths ? a.seq_lengths[b] : F;
  const size_t row = (size_t)b * F + t;
  LstmCellBwdPre p;
  p.live = t < len;
  p.dc_in = a.dc[idx];
  const float* g4 = a.gates + row * 4 * H + j;
  p.gi = g4[0]; p.gf = g4[H]; p.gg = g4[2 * H]; p.go = g4[3 * H];
  p.tc = tanhf(a.c_all[row * H + j]);
  p.c_prev = t > 0 ? a.c_all[(row - 1) * H + j] : (a.c0 ? a.c0[idx] : 0.f);
  p.dy = a.dy ? a.dy[row * a.ld_dy + j] : 0.f;
  return p;
}
__device__ __forceinline__ void lstm_cell_bwd_finish(const LstmCellBwdArgs& a, int idx, const LstmCellBwdPre& p, float dh_in) {
  const int b = idx / a.H, j = idx - b * a.H;
  const int H = a.H;
  float* dG = a.dgates + ((size_t)b * a.F + a.t) * 4 * H + j;
  if (!p.live) {
    dG[0] = 0.f; dG[H] = 0.f; dG[2 * H] = 0.f; dG[3 * H] = 0.f;
    a.dh_carry[idx] = dh_in;
    return;   // dc passes through unchanged
  }
  const float dh = p.dy + dh_in;
  const float d_o = dh * p.tc;
  const float dc = p.dc_in + dh * p.go * (1.f - p.tc * p.tc);
  dG[0] = dc * p.gg * p.gi * (1.f - p.gi);
  dG[H] = dc * p.c_prev * p.gf * (1.f - p.gf);
  dG[2 * H] = dc * p.gi * (1.f - p.gg * p.gg);
  dG[3 * H] = d_o * p.go * (1.f - p.go);
  a.dc[idx] = dc * p.gf;
  a.dh_carry[idx] = 0.f;
}

__device__ __forceinline__ void lstm_cell_bwd_elem(const LstmCellBwdArgs& a, int idx, float dh_in) {
  const int b = idx / a.H, j = idx - b * a.H;
  const int H = a.H, F = a.F, t = a.t;
  const int len = a.seq_lengths ? a.seq_lengths[b] : F;
  const size_t row = (size_t)b * F + t;
  float* dG = a.dgates + row * 4 * H + j;
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

}  // namespace empose
