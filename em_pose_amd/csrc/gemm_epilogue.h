// Shared by the fp32 matrix-core GEMM kernels (gemm_f32.hip, mlp_fused.hip): accumulator vector types and the tile
// epilogue. Internal, gfx950 only.
#pragma once
#include "kernels.h"

namespace empose {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Epilogue of a wave's WM x WN grid of 32x32 accumulator tiles whose top-left element is (mw0, nw0): per-column
// scale/shift (bias, folded eval-mode BatchNorm), activation, residual.  C/D layout of the 32x32 MFMA:
// col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
// Everything the loops need is copied out of the (kernarg-resident) problem descriptor first and the activation /
// residual / edge cases are resolved OUTSIDE the loops: the stores through p.C could alias the descriptor as far as
// the compiler knows, which otherwise costs a scalar reload + wait per stored element.
// Addresses are a wave-uniform base (scalar registers) plus a 32-bit per-lane byte offset, so an element costs one
// integer add besides its arithmetic and its store (64-bit per-lane address arithmetic doubled the epilogue).
typedef __attribute__((address_space(1))) char* epi_gbyte_t;
typedef const __attribute__((address_space(1))) char* epi_cgbyte_t;
typedef __attribute__((address_space(1))) float* epi_gfloat_t;
typedef const __attribute__((address_space(1))) float* epi_cgfloat_t;

template <int WM, int WN, int MODE, bool FULL>   // MODE 0: act 0/1, 1: act 0/1 + residual, 2: residual block (act 2)
__device__ __forceinline__ void epilogue_mode(float* __restrict__ C, const float* __restrict__ resid, long ldc, long ldr,
                                              int M, int N, float slope, const float* __restrict__ scale,
                                              const float* __restrict__ shift, const f32x16 (&acc)[WM][WN], int mw0,
                                              int nw0, int l31, int lh) {
  epi_gbyte_t cb = (epi_gbyte_t)(C + (long)mw0 * ldc);                       // the wave's first row
  epi_cgbyte_t rb = MODE ? (epi_cgbyte_t)(resid + (long)mw0 * ldr) : nullptr;
  const unsigned ldc4 = (unsigned)ldc * 4u, ldr4 = (unsigned)ldr * 4u;
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const int n = nw0 + j * 32 + l31;
    if (n >= N) continue;
    const float sc = scale ? scale[n] : 1.f;
    const float sh = shift ? shift[n] : 0.f;
    const unsigned c_lane = (unsigned)(4 * lh) * ldc4 + (unsigned)n * 4u;
    const unsigned r_lane = (unsigned)(4 * lh) * ldr4 + (unsigned)n * 4u;
#pragma unroll
    for (int i = 0; i < WM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dm = i * 32 + (r & 3) + 8 * (r >> 2);
        if (!FULL && mw0 + 4 * lh + dm >= M) continue;
        float y = acc[i][j][r] * sc + sh;
        if (MODE == 2) {        // relu(W x + b + x), reference layers.py:170-182
          if (rb) y += *(epi_cgfloat_t)(rb + (r_lane + (unsigned)dm * ldr4));
          y = y > 0.f ? y : 0.f;
        } else {
          y = y >= 0.f ? y : slope * y;      // slope == 1 when there is no activation (exact identity)
          if (MODE == 1) y += *(epi_cgfloat_t)(rb + (r_lane + (unsigned)dm * ldr4));  // skip connection (after the act.)
        }
        *(epi_gfloat_t)(cb + (c_lane + (unsigned)dm * ldc4)) = y;
      }
    }
  }
}

template <int WM, int WN>
__device__ __forceinline__ void epilogue(const GemmProb& p, const f32x16 (&acc)[WM][WN], int mw0, int nw0, int l31,
                                         int lh) {
  float* C = p.C;
  const float* resid = p.resid;
  const float* scale = p.scale;
  const float* shift = p.shift;
  const long ldc = p.ldc, ldr = p.ldr;
  const int M = p.M, N = p.N, act = p.act;
  const float slope = act == 1 ? p.slope : 1.f;
  mw0 = __builtin_amdgcn_readfirstlane(mw0);   // wave-uniform by construction; tell the compiler
  nw0 = __builtin_amdgcn_readfirstlane(nw0);
  const bool full = mw0 + 32 * WM <= M;
#define EMPOSE_EPI(MODE, FULL) \
  epilogue_mode<WM, WN, MODE, FULL>(C, resid, ldc, ldr, M, N, slope, scale, shift, acc, mw0, nw0, l31, lh)
  if (act == 2) { if (full) EMPOSE_EPI(2, true); else EMPOSE_EPI(2, false); }
  else if (resid) { if (full) EMPOSE_EPI(1, true); else EMPOSE_EPI(1, false); }
  else { if (full) EMPOSE_EPI(0, true); else EMPOSE_EPI(0, false); }
#undef EMPOSE_EPI
}

}  // namespace empose
