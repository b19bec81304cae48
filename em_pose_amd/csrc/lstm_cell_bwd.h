// One element (batch row b, hidden unit j) of a step of back-propagation through time of an LSTM layer (see the comment of
// lstm_cell_bwd_kernel in train.hip).  Shared by that kernel and by the reduce kernel of the K-split recurrent product
// (gemm_f32.hip), which feeds the dh it has just formed straight into the next step's cell instead of storing it.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace empose {

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
