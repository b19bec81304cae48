// The fused update MLP (mlp_fused.hip: all layers of one or two nets in ONE launch, activations resident in LDS) with
// every fp32 product formed on the bf16 matrix path from THREE bf16 pieces per operand -- fp32-equivalent arithmetic at
// 6/16 of the fp32 MFMA's cost.
//
//   x = x_h + x_m + x_l,   x_h = bf16(x), x_m = bf16(x - x_h), x_l = bf16(x - x_h - x_m)      (round to nearest, 8 + 8 + 8
//   w = w_h + w_m + w_l                                                                        mantissa bits: all 24)
//   x w  ~=  x_h w_h + (x_h w_m + x_m w_h) + (x_h w_l + x_m w_m + x_l w_h)
// Each piece product is exact (8 x 8 bits), the accumulation is the matrix core's fp32; what is dropped -- x_m w_l,
// x_l w_m, x_l w_l -- is below 2^-23 of the product, i.e. below the rounding the fp32 instruction commits on its own
// products.  (Two pieces, 2^-16, would not do for the 1e-4 parity bar through 4 x 6 layers; this is NOT the opt-in
// "bf16x3" of the full-mesh kernel, which keeps two pieces.)
// v_mfma_f32_32x32x16_bf16 retires 16 k per 32 cycles and SIMD where v_mfma_f32_32x32x2_f32 retires 2 k per 64: six of
// them per 16 k = 192 cycles against 512.
//
// Structure = mlp_fused.hip: a workgroup owns 64 rows of one net, its activations [64][<= 512] stay in ONE fp32 LDS
// buffer (three bf16 planes of it, 196 KB, would not fit), four waves of 64 rows x 128 columns, two barriers per layer.
//   * A side: a lane reads its 8 consecutive fp32 of a k-step (two ds_read_b128) and splits them into the three pieces in
//     registers (v_cvt_pk_bf16_f32 + shift/mask + subtract: 11 VALU per pair, 88 per k-step for the two row tiles,
//     hidden under the 48 MFMAs of the step).  Every wave splits the same A block: redundant, but free in the MFMA shadow,
//     whereas split planes in LDS would cost 64 more KB.
//   * B side: the weights are split ONCE at model creation (api.hip pack_fragments_x3) and stored in fragment order, per
//     (k-step, 32-column tile, piece) one 1 KB wave fragment: lane (n = lane & 31, half = lane >> 5) owns
//     W_piece[tile * 32 + n][ks * 16 + half * 8 .. + 7].  They stream from L2 straight into a register ring, three
//     k-steps deep; 12 KB per wave and k-step.
//   * one wave per SIMD (132 KB of LDS per workgroup), as everything in this tree that issues the bf16 32x32x16 MFMA
//     (scripts/dev/bf16_hazard_repro.md).
#include "bf16x3.h"
#include "gemm_epilogue.h"

namespace empose {

namespace fx {
constexpr int BM = 64, NT = 256;
constexpr int LDA = FUSED_MAX_WIDTH + 4;
constexpr size_t LDS_BYTES = (size_t)BM * LDA * sizeof(float) + 64;
constexpr int RING = 4;                      // weight ring: three k-steps in flight + the one being multiplied
constexpr int SG_MFMA = 0x008, SG_VALU = 0x002, SG_VMEM_RD = 0x020, SG_DS_RD = 0x100;
}  // namespace fx

typedef const __attribute__((address_space(1))) u32x4_t* fx_gvec_t;
typedef const __attribute__((address_space(1))) char* fx_gbyte_t;

#define FX_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)

// the pieces of a lane's 8 consecutive k
__device__ __forceinline__ Pieces fx_split8(const f32x4& lo, const f32x4& hi) {
#ifdef FX_LAB_NOSPLIT   // dev lab: no arithmetic on the A side (what the loop costs without the split)
  Pieces z;
  z.p[0] = __builtin_bit_cast(u32x4_t, lo); z.p[1] = __builtin_bit_cast(u32x4_t, hi); z.p[2] = z.p[0];
  return z;
#endif
  return split8(lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]);
}

// One layer for the workgroup's 64 rows.  `act`: fp32 [64][lda], columns [0, 16 * KS4) valid or zero.
// The wave computes WM x WN tiles of 32 x 32 starting at (row_tile0, col_tile0).
template <int WM, int WN>
__device__ __forceinline__ void x3_layer(const FusedNet& net, const FusedLayer& L, int M, int m0, float* act, int lda,
                                         int row_tile0, int col_tile0, bool last) {
  using namespace fx;
  const int lane = threadIdx.x & 63;
  const int l31 = lane & 31, lh = lane >> 5;
  const int K = L.K, N = L.N;
  const int NT32 = (N + 31) / 32;
  const int KS4 = ((K + 15) / 16 + 3) & ~3;   // k-steps of the packed weights (zero-padded to whole quads)
  if (col_tile0 >= NT32) {   // wave-uniform: nothing of this layer falls to this wave; keep the two barriers
    __syncthreads();
    __syncthreads();
    return;
  }
  unsigned b_voff[WN];
#pragma unroll
  for (int j = 0; j < WN; ++j)
    b_voff[j] = (unsigned)((col_tile0 + j < NT32 ? col_tile0 + j : NT32 - 1) * 3072 + lane * 16);
  fx_gbyte_t wb = (fx_gbyte_t)L.W;
  const float* a_rd = act + (row_tile0 * 32 + l31) * lda + lh * 8;

  float e_sc[WN], e_sh[WN];
  const float e_slope = L.act == 1 ? L.slope : 1.f;
  const bool slope_unit = e_slope >= 0.f && e_slope <= 1.f;
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const int n = (col_tile0 + j) * 32 + l31;
    const bool real = n < N;
    const int nc = real ? n : N - 1;
    e_sc[j] = real ? (L.scale ? L.scale[nc] : 1.f) : 0.f;
    e_sh[j] = real ? (L.shift ? L.shift[nc] : 0.f) : 0.f;
  }

  f32x16 acc[WM][WN];
  f32x4 ra[2][WM][2];              // raw fp32 A: k-steps s + 1 (being split) and s + 2 (in flight from LDS)
  Pieces ap[2][WM];                // the pieces of k-steps s (multiplied) and s + 1 (being made)
  u32x4_t fb[RING][WN][3];         // weight pieces: slot s & 3 holds k-step s, loaded three steps ahead

  auto aread = [&](int ks, f32x4 (&a)[WM][2]) {
    const int kc = ks < KS4 ? ks : KS4 - 1;   // (the look-ahead past the last step stays inside the padded row)
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      a[i][0] = *reinterpret_cast<const f32x4*>(a_rd + i * 32 * lda + kc * 16);
      a[i][1] = *reinterpret_cast<const f32x4*>(a_rd + i * 32 * lda + kc * 16 + 4);
    }
  };
  auto bload = [&](u32x4_t (&b)[WN][3], int ks) {
#ifdef FX_LAB_NOB   // dev lab: only the prologue's weight loads (what the loop costs without its weight stream)
    if (ks >= 3) return;
#endif
#ifdef FX_LAB_SAMEB   // dev lab: every k-step fetches the weights of step 0 (the loads' issue cost without their stream)
    const int kc = 0 * ks;
#else
    const int kc = ks < KS4 ? ks : KS4 - 1;   // (clamped: fetched, never used)
#endif
    fx_gbyte_t p = wb + (size_t)kc * NT32 * 3072;
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int q = 0; q < 3; ++q) b[j][q] = *(fx_gvec_t)(p + b_voff[j] + q * 1024);
  };
  auto split = [&](const f32x4 (&a)[WM][2], Pieces (&q)[WM]) {
#pragma unroll
    for (int i = 0; i < WM; ++i) q[i] = fx_split8(a[i][0], a[i][1]);
  };
  // the six products of a k-step, the small ones first so that they meet in the accumulator before the large one rounds;
  // consecutive MFMAs go to different accumulator tiles
  auto mma = [&](const Pieces (&a)[WM], const u32x4_t (&b)[WN][3]) {
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[i].p[X3_PA[t]]),
                                                              __builtin_bit_cast(bf16x8_t, b[j][X3_PB[t]]), acc[i][j], 0, 0, 0);
  };
  // one k-step: its MFMAs with the next step's split (VALU), the weight loads of step s + 3 and the LDS reads of step s + 2
  // spread between them
  auto pattern = [&]() {
    constexpr int NM = WM * WN * 6, NV = 48 * WM, NVM = 3 * WN, NDS = 2 * WM;
    constexpr int vper = (NV + NM - 1) / NM;
    static_assert(NVM + NDS <= NM, "k-step too small for its memory operations");
#pragma unroll
    for (int q = 0; q < NM; ++q) {
      FX_SGB(SG_MFMA, 1);
      if (q < NVM) FX_SGB(SG_VMEM_RD, 1);
      else if (q < NVM + NDS) FX_SGB(SG_DS_RD, 1);
      FX_SGB(SG_VALU, vper);
    }
  };
  // One k-step as NCH chunks, each fenced by a scheduling barrier: 48 / NCH of the step's MFMAs, the split of ONE pair of the
  // next step's A values (11 VALU) and 16 / NCH of the step's memory operations (the weight loads of step s + 3, the LDS
  // reads of step s + 2).  Left to itself over a whole step (or four), the scheduler emits the split as one dependent
  // chain behind the MFMAs and sinks the LDS reads down to their use, where the split then waits for the round trip.
  auto step = [&](int s, const Pieces (&ap_cur)[WM], Pieces (&ap_nxt)[WM], f32x4 (&ra_nxt)[WM][2], f32x4 (&ra_free)[WM][2],
                  const u32x4_t (&b_cur)[WN][3], u32x4_t (&b_free)[WN][3]) {
    constexpr int NM = WM * WN * 6, NP = WM * 4, NMEM = 3 * WN + 2 * WM;   // MFMAs, pairs to split, memory operations
    constexpr int NCH = NP;                                                // one pair per chunk
    const int kb = s + 3 < KS4 ? s + 3 : KS4 - 1, ka = s + 2 < KS4 ? s + 2 : KS4 - 1;
    fx_gbyte_t pb = wb + (size_t)kb * NT32 * 3072;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      // ---- memory operations of this chunk
#pragma unroll
      for (int o = c * NMEM / NCH; o < (c + 1) * NMEM / NCH; ++o) {
        if (o < 3 * WN) {
#ifndef FX_LAB_NOB
          b_free[o / 3][o % 3] = *(fx_gvec_t)(pb + b_voff[o / 3] + (o % 3) * 1024);
#endif
        } else {
          const int r = o - 3 * WN;
          ra_free[r / 2][r % 2] = *reinterpret_cast<const f32x4*>(a_rd + (r / 2) * 32 * lda + ka * 16 + (r % 2) * 4);
        }
      }
      // ---- the split of pair c of the next step: row tile c / 4, k pair c % 4
      {
        const int i = c / 4, q = c % 4;
        const f32x4& v = ra_nxt[i][q / 2];
        unsigned h, m, l;
#ifdef FX_LAB_NOSPLIT
        h = __builtin_bit_cast(unsigned, v[(q % 2) * 2]); m = __builtin_bit_cast(unsigned, v[(q % 2) * 2 + 1]); l = h;
#else
        split_pair(v[(q % 2) * 2], v[(q % 2) * 2 + 1], h, m, l);
#endif
        ap_nxt[i].p[0][q] = h; ap_nxt[i].p[1][q] = m; ap_nxt[i].p[2][q] = l;
      }
      // ---- this chunk's share of the step's MFMAs, in (product, row tile, column tile) order
#pragma unroll
      for (int mm = c * NM / NCH; mm < (c + 1) * NM / NCH; ++mm) {
        const int t = mm / (WM * WN), i = (mm / WN) % WM, j = mm % WN;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ap_cur[i].p[X3_PA[t]]),
                                                            __builtin_bit_cast(bf16x8_t, b_cur[j][X3_PB[t]]), acc[i][j], 0, 0, 0);
      }
#pragma unroll
      for (int mm = 0; mm < (NM + NCH - 1) / NCH; ++mm) { FX_SGB(SG_MFMA, 1); FX_SGB(SG_VALU, 2); if (mm < 2) FX_SGB(SG_VMEM_RD | SG_DS_RD, 1); }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto quad = [&](int g) {
    step(g, ap[0], ap[1], ra[1], ra[0], fb[0], fb[3]);
    step(g + 1, ap[1], ap[0], ra[0], ra[1], fb[1], fb[0]);
    step(g + 2, ap[0], ap[1], ra[1], ra[0], fb[2], fb[1]);
    step(g + 3, ap[1], ap[0], ra[0], ra[1], fb[3], fb[2]);
  };

#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  bload(fb[0], 0);
  bload(fb[1], 1);
  bload(fb[2], 2);
  aread(0, ra[0]);
  aread(1, ra[1]);
  split(ra[0], ap[0]);
  // the first quad is peeled so that the loop header merges two states with the same outstanding loads (mlp_fused.hip)
  quad(0);
  for (int g = 4; g < KS4; g += 4) quad(g);

  __syncthreads();   // every wave has read its last A fragment: the buffer may be overwritten
  if (last) {
    GemmProb p;
    p.C = net.out; p.ldc = net.ld_out;
    p.M = M; p.N = N; p.K = K;
    p.scale = L.scale; p.shift = L.shift; p.resid = nullptr; p.ldr = 0;
    p.act = L.act; p.slope = L.slope;
    p.A = nullptr; p.W = nullptr; p.lda = 0; p.ldw = 0;
    epilogue<WM, WN>(p, acc, m0 + row_tile0 * 32, col_tile0 * 32, l31, lh);
  } else {
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int n = (col_tile0 + j) * 32 + l31;
      if (col_tile0 + j >= NT32) continue;
      if (slope_unit) {
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = (row_tile0 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const float y = acc[i][j][r] * e_sc[j] + e_sh[j];
            act[row * lda + n] = fmaxf(y, y * e_slope);
          }
      } else {
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = (row_tile0 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const float y = acc[i][j][r] * e_sc[j] + e_sh[j];
            act[row * lda + n] = y >= 0.f ? y : y * e_slope;
          }
      }
    }
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------------
// The same layer with the A-side split done ONCE per element instead of once per wave: the four waves each split a quarter
// of the k-step's 64 x 16 block (16 rows: one fp32 ds_read_b128 and 22 VALU per lane instead of 88) and leave the pieces
// in a small LDS stage in FRAGMENT order ([piece][row tile] -> 1 KB, so a wave's A fragment is one conflict-free
// ds_read_b128); one workgroup barrier per k-step hands them over.  Pieces of step k: raw read at the start of step k - 2,
// split + written at its end, barrier at the start of step k - 1, fragments read right after it, multiplied in step k.
// Two stages of 6 KB behind the activation buffer.  Every wave runs the loop, also one that owns no column of a narrow
// layer (it still splits its rows and meets the barriers).
// MEASURED (T = 32768, two nets, scripts/dev/fused_x3_lab.hip): 853-879 us per launch against 780-798 us for the kernel above
// -- the 66 VALU per lane and k-step it saves cost less than the barrier per k-step that replaces them (the four waves then
// run in lock step: every stall of one is a stall of all; split early or late in the step, raw values read one or two steps
// ahead: within 3 %).  Opt-in (option mlp_x3 = 2); kept for what it shows.
namespace fx {
constexpr int STAGE_U32 = 3 * 2 * 64 * 4;                          // one k-step's pieces: [piece][row tile][lane][4 x u32]
constexpr int ACT_FLOATS = BM * LDA + 16;
constexpr size_t LDS_BYTES_C = ((size_t)ACT_FLOATS + 2 * STAGE_U32) * sizeof(float);   // 144,448 bytes
}  // namespace fx

template <int WM, int WN>
__device__ __forceinline__ void x3c_layer(const FusedNet& net, const FusedLayer& L, int M, int m0, float* act, int lda,
                                          int row_tile0, int col_tile0, bool last) {
  using namespace fx;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int K = L.K, N = L.N;
  const int NT32 = (N + 31) / 32;
  const int KS4 = ((K + 15) / 16 + 3) & ~3;
  const bool active = col_tile0 < NT32;     // wave-uniform
  unsigned* stage = reinterpret_cast<unsigned*>(act + ACT_FLOATS);
  unsigned b_voff[WN];
#pragma unroll
  for (int j = 0; j < WN; ++j)
    b_voff[j] = (unsigned)((col_tile0 + j < NT32 ? col_tile0 + j : NT32 - 1) * 3072 + lane * 16);
  fx_gbyte_t wb = (fx_gbyte_t)L.W;
  // producer side: this lane's four raw values of a k-step and where their pieces go
  const int p_row = wave * 16 + (lane >> 2), p_k4 = (lane & 3) * 4;
  const float* p_rd = act + p_row * lda + p_k4;
  const int p_wr = ((p_row >> 5) * 64 + (p_row & 31) + 32 * (p_k4 >> 3)) * 4 + ((p_k4 >> 2) & 1) * 2;   // u32 index in a piece plane
  // consumer side: fragment (piece pc, row tile i) of a stage = 64 lanes x 16 bytes
  const int c_rd = (row_tile0 * 64 + lane) * 4;

  float e_sc[WN], e_sh[WN];
  const float e_slope = L.act == 1 ? L.slope : 1.f;
  const bool slope_unit = e_slope >= 0.f && e_slope <= 1.f;
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const int n = (col_tile0 + j) * 32 + l31;
    const bool real = n < N;
    const int nc = real ? n : N - 1;
    e_sc[j] = real ? (L.scale ? L.scale[nc] : 1.f) : 0.f;
    e_sh[j] = real ? (L.shift ? L.shift[nc] : 0.f) : 0.f;
  }

  f32x16 acc[WM][WN];
  f32x4 raw[2];                    // this lane's raw values of k-steps s + 2 (split in step s) and s + 3: read TWO steps
                                   // before their split, so that the split has its data at the start of the step
  Pieces ap[2][WM];                // fragments of k-steps s (multiplied) and s + 1 (in flight from the stage)
  u32x4_t fb[RING][WN][3];

  auto rawread = [&](int ks, f32x4& r) {
    const int kc = ks < KS4 ? ks : KS4 - 1;
    r = *reinterpret_cast<const f32x4*>(p_rd + kc * 16);
  };
  auto produce = [&](int ks, const f32x4& r) {     // split r (the raw values of k-step ks) into stage ks & 1
    unsigned h0, m0_, l0, h1, m1, l1;
    split_pair(r[0], r[1], h0, m0_, l0);
    split_pair(r[2], r[3], h1, m1, l1);
    unsigned* st = stage + (ks & 1) * STAGE_U32 + p_wr;
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    *reinterpret_cast<u32x2_t*>(st) = u32x2_t{h0, h1};
    *reinterpret_cast<u32x2_t*>(st + 2 * 64 * 4) = u32x2_t{m0_, m1};
    *reinterpret_cast<u32x2_t*>(st + 2 * 2 * 64 * 4) = u32x2_t{l0, l1};
  };
  auto fetch = [&](int ks, Pieces (&q)[WM]) {   // fragments of k-step ks out of its stage
    const unsigned* st = stage + (ks & 1) * STAGE_U32 + c_rd;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) q[i].p[pc] = *reinterpret_cast<const u32x4_t*>(st + (pc * 2 + i) * 64 * 4);
  };
  auto bload = [&](u32x4_t (&b)[WN][3], int ks) {
    const int kc = ks < KS4 ? ks : KS4 - 1;
    fx_gbyte_t p = wb + (size_t)kc * NT32 * 3072;
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int q = 0; q < 3; ++q) b[j][q] = *(fx_gvec_t)(p + b_voff[j] + q * 1024);
  };
  auto mma = [&](const Pieces (&a)[WM], const u32x4_t (&b)[WN][3]) {
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[i].p[X3_PA[t]]),
                                                              __builtin_bit_cast(bf16x8_t, b[j][X3_PB[t]]), acc[i][j], 0, 0, 0);
  };
  auto pattern = [&]() {
    constexpr int NM = WM * WN * 6, NVM = 3 * WN, NDS = 3 * WM + 1;
    if constexpr (NM >= 48) {      // (the narrow output layers are left to the compiler's own order)
      // the split of step s + 2 and its three LDS writes FIRST (they must have landed long before the next barrier, which
      // waits for this wave's LDS operations), then the fragment reads, then the weight loads
#pragma unroll
      for (int q = 0; q < 12; ++q) { FX_SGB(SG_MFMA, 1); FX_SGB(SG_VALU, 2); }
#pragma unroll
      for (int q = 0; q < 3; ++q) { FX_SGB(SG_MFMA, 1); FX_SGB(0x200, 1); }
#pragma unroll
      for (int q = 0; q < NDS; ++q) { FX_SGB(SG_MFMA, 1); FX_SGB(SG_DS_RD, 1); }
#pragma unroll
      for (int q = 0; q < NVM; ++q) { FX_SGB(SG_MFMA, 1); FX_SGB(SG_VMEM_RD, 1); }
      FX_SGB(SG_MFMA, NM - 15 - NDS - NVM);
    }
  };
  // step s: barrier (stage (s + 1) & 1 is complete) | split + write of s + 2 | fragments of s + 1, raw of s + 3 | weights of
  // s + 3 | MFMAs of s
  auto step = [&](int s, const Pieces (&ap_cur)[WM], Pieces (&ap_nxt)[WM], const u32x4_t (&b_cur)[WN][3],
                  u32x4_t (&b_free)[WN][3], f32x4& r) {
    __syncthreads();
    produce(s + 2, r);             // (r holds k-step s + 2, read during step s - 2; stage s & 1 was last read in step s - 1)
    fetch(s + 1, ap_nxt);
    rawread(s + 4, r);
    if (active) {
      bload(b_free, s + 3);
      mma(ap_cur, b_cur);
      pattern();
    }
  };
  auto quad = [&](int g) {
    step(g, ap[0], ap[1], fb[0], fb[3], raw[0]);
    step(g + 1, ap[1], ap[0], fb[1], fb[0], raw[1]);
    step(g + 2, ap[0], ap[1], fb[2], fb[1], raw[0]);
    step(g + 3, ap[1], ap[0], fb[3], fb[2], raw[1]);
  };

#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- prologue: pieces of steps 0 and 1 into their stages, fragments of step 0 into registers, raw of steps 2, 3 in flight
  if (active) {
    bload(fb[0], 0);
    bload(fb[1], 1);
    bload(fb[2], 2);
  }
  rawread(0, raw[0]);
  rawread(1, raw[1]);
  produce(0, raw[0]);
  produce(1, raw[1]);
  rawread(2, raw[0]);
  rawread(3, raw[1]);
  __syncthreads();
  fetch(0, ap[0]);
  quad(0);
  for (int g = 4; g < KS4; g += 4) quad(g);

  __syncthreads();   // every wave has read its last raw values and fragments: the buffer may be overwritten
  if (active) {
    if (last) {
      GemmProb p;
      p.C = net.out; p.ldc = net.ld_out;
      p.M = M; p.N = N; p.K = K;
      p.scale = L.scale; p.shift = L.shift; p.resid = nullptr; p.ldr = 0;
      p.act = L.act; p.slope = L.slope;
      p.A = nullptr; p.W = nullptr; p.lda = 0; p.ldw = 0;
      epilogue<WM, WN>(p, acc, m0 + row_tile0 * 32, col_tile0 * 32, l31, lh);
    } else {
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int n = (col_tile0 + j) * 32 + l31;
        if (col_tile0 + j >= NT32) continue;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = (row_tile0 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const float y = acc[i][j][r] * e_sc[j] + e_sh[j];
            act[row * lda + n] = slope_unit ? fmaxf(y, y * e_slope) : (y >= 0.f ? y : y * e_slope);
          }
      }
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(fx::NT) void mlp_fused_x3c_kernel(FusedMlpArgs args) {
  X3_EXCLUSIVE_SIMD();
  using namespace fx;
  extern __shared__ __attribute__((aligned(16))) float act[];
  const FusedNet& net = args.net[blockIdx.y];
  const int M = args.M, m0 = blockIdx.x * BM;
  const int tid = threadIdx.x;
  {
    const int K0 = net.layer[0].K;
    const int kpad = (K0 + 63) / 64 * 64;
    const int c4n = kpad / 4;
    for (int i = tid; i < BM * c4n; i += NT) {
      const int r = i / c4n, c = (i % c4n) * 4;
      const int row = m0 + r < M ? m0 + r : M - 1;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (c < K0) v = *reinterpret_cast<const f32x4*>(net.x + (size_t)row * net.ldx + c);
      *reinterpret_cast<f32x4*>(act + r * LDA + c) = v;
    }
  }
  __syncthreads();
  for (int l = 0; l < net.n_layers; ++l) {
    const FusedLayer& L = net.layer[l];
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool last = l == net.n_layers - 1;
    if (L.N <= 32) x3c_layer<1, 1>(net, L, M, m0, act, LDA, wave & 1, wave >> 1, last);
    else if (L.N <= 128) x3c_layer<1, 2>(net, L, M, m0, act, LDA, wave & 1, (wave >> 1) * 2, last);
    else x3c_layer<2, 4>(net, L, M, m0, act, LDA, 0, wave * 4, last);
  }
}

__global__ __launch_bounds__(fx::NT) void mlp_fused_x3_kernel(FusedMlpArgs args) {
  X3_EXCLUSIVE_SIMD();
  using namespace fx;
  extern __shared__ __attribute__((aligned(16))) float act[];
  const FusedNet& net = args.net[blockIdx.y];
  const int M = args.M, m0 = blockIdx.x * BM;
  const int tid = threadIdx.x;
  {
    const int K0 = net.layer[0].K;
    const int kpad = (K0 + 63) / 64 * 64;   // whole quads of k-steps
    const int c4n = kpad / 4;
    for (int i = tid; i < BM * c4n; i += NT) {
      const int r = i / c4n, c = (i % c4n) * 4;
      const int row = m0 + r < M ? m0 + r : M - 1;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (c < K0) v = *reinterpret_cast<const f32x4*>(net.x + (size_t)row * net.ldx + c);   // K0 % 4 == 0
      *reinterpret_cast<f32x4*>(act + r * LDA + c) = v;
    }
  }
  __syncthreads();
  for (int l = 0; l < net.n_layers; ++l) {
    const FusedLayer& L = net.layer[l];
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool last = l == net.n_layers - 1;
    if (L.N <= 32) x3_layer<1, 1>(net, L, M, m0, act, LDA, wave & 1, wave >> 1, last);
    else if (L.N <= 128) x3_layer<1, 2>(net, L, M, m0, act, LDA, wave & 1, (wave >> 1) * 2, last);
    else x3_layer<2, 4>(net, L, M, m0, act, LDA, 0, wave * 4, last);
  }
}

hipError_t launch_mlp_fused_x3(const FusedMlpArgs& args, hipStream_t stream) {
  dim3 grid((args.M + fx::BM - 1) / fx::BM, args.count);
  if (options().mlp_x3 >= 2) {   // opt-in: the cooperative split (measured slower, see x3c_layer)
    if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(mlp_fused_x3c_kernel), fx::LDS_BYTES_C)) return e;
    hipLaunchKernelGGL(mlp_fused_x3c_kernel, grid, dim3(fx::NT), fx::LDS_BYTES_C, stream, args);
    return hipGetLastError();
  }
  if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(mlp_fused_x3_kernel), fx::LDS_BYTES)) return e;
  hipLaunchKernelGGL(mlp_fused_x3_kernel, grid, dim3(fx::NT), fx::LDS_BYTES, stream, args);
  return hipGetLastError();
}

}  // namespace empose
