// fp32 values as THREE bf16 pieces (x = h + m + l; every piece the round-to-nearest bf16 of what the previous ones leave:
// 8 + 8 + 8 mantissa bits = all 24 of an fp32), the operand format of the kernels that form fp32 products on the bf16
// matrix path (mlp_fused_x3.hip, lstm_x3.hip):
//   x w ~= x_h w_h + (x_h w_m + x_m w_h) + (x_h w_l + x_m w_m + x_l w_h)        six v_mfma_f32_32x32x16_bf16 products;
// what is dropped (x_m w_l, x_l w_m, x_l w_l) is below 2^-23 of the product.  Internal, gfx950 only.
#pragma once
#include "kernels.h"

namespace empose {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// two fp32 -> their three bf16 pieces, packed (element 0 in the low half): v_cvt_pk_bf16_f32 + shift / mask + subtract
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  h = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{x0, x1}, bf16x2_t));
  const float r0 = x0 - __builtin_bit_cast(float, h << 16);
  const float r1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
  m = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{r0, r1}, bf16x2_t));
  const float s0 = r0 - __builtin_bit_cast(float, m << 16);
  const float s1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{s0, s1}, bf16x2_t));
}

struct Pieces { u32x4_t p[3]; };   // (h, m, l) of 8 consecutive k of one row: three A / B fragments of the 32x32x16 MFMA

__device__ __forceinline__ Pieces split8(float x0, float x1, float x2, float x3, float x4, float x5, float x6, float x7) {
  unsigned h[4], m[4], l[4];
  split_pair(x0, x1, h[0], m[0], l[0]);
  split_pair(x2, x3, h[1], m[1], l[1]);
  split_pair(x4, x5, h[2], m[2], l[2]);
  split_pair(x6, x7, h[3], m[3], l[3]);
  Pieces q;
  q.p[0] = u32x4_t{h[0], h[1], h[2], h[3]};
  q.p[1] = u32x4_t{m[0], m[1], m[2], m[3]};
  q.p[2] = u32x4_t{l[0], l[1], l[2], l[3]};
  return q;
}

// A wave that issues v_mfma_f32_32x32x16_bf16 must have its SIMD to itself: a global store issued by ANY wave of the SIMD
// while the instruction is in flight corrupts an accumulator element (scripts/dev/bf16_hazard_repro.md).  One workgroup per
// CU by its LDS request keeps this kernel's own workgroups apart, but not the workgroups of another stream's kernel that
// needs no LDS (the training engine's side streams, the streaming evaluation driver, another process).  Touching the last
// architectural and the last accumulation register makes the wave allocate all 512 registers of its SIMD lane: no other
// wave fits beside it, whatever it is -- exclusivity by construction (round 6).  First statement of every such kernel.
#ifdef X3_LAB_SHARED_SIMD      // (scripts/dev/x3_shared_simd_lab.sh: the library WITHOUT the guarantee, to show what it is for)
#define X3_EXCLUSIVE_SIMD() do { } while (0)
#else
#define X3_EXCLUSIVE_SIMD() asm volatile("v_mov_b32 v255, 0\n\tv_accvgpr_write_b32 a255, 0" ::: "v255", "a255")
#endif

// The six piece products of a k-step in the order they are issued: the small ones first, so that they meet in the
// accumulator before the large one rounds.
constexpr int X3_PA[6] = {2, 1, 0, 1, 0, 0}, X3_PB[6] = {0, 1, 2, 0, 1, 0};

}  // namespace empose
