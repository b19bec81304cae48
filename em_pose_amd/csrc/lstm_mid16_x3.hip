// The LSTM wavefront step of batches of 9 .. 64 rows on ALL 256 CUs (round 6): lstm_mid_x3.hip with half the tile.
//
// lstm_mid_x3_kernel gives a workgroup 8 hidden units x 4 gates = one 32-column tile: 64 workgroups per layer, 128 for the
// two layers of a wavefront step -- half of the chip -- and each streams 196 KB of weight pieces per step, which is what ONE
// CU can pull from the memory side in the 3.6 us its K loop takes (the L2 does not keep weights across launches).  With
// twice the workgroups and half the K each the same launch takes 7.2 instead of 8.2 us at 32 rows and 8.2 instead of 10.1
// at 36 (lab build LM3_LAB_TWICE_WGS, profiles/r06i_lstm_mid_ring_lab.txt).  Hence this form:
//   * a workgroup owns 4 hidden units x 4 gates = 16 columns (column n = gate * 4 + unit) of up to 64 rows: 128 workgroups
//     per layer, 256 per step; weights packed [k-step of 32][4-unit block][piece] (api.hip pack_lstm_x3_mid16);
//   * the products are v_mfma_f32_16x16x32_bf16: 16 rows x 16 columns x 32 k.  The A planes stay as every other kernel
//     writes them ([32-row tile][k-step of 16][piece][lane (row, k half)][8]): lane (row r, k quarter q) of a 16 x 32
//     fragment reads the 16 bytes of row r, half q & 1 of k-step 2 K + (q >> 1) -- 16 lanes a contiguous 256 bytes;
//   * four waves split the k-steps of 32 (wave w: w, w + 4, ...), ring of D steps in flight, partial sums meet in LDS,
//     thread (row, unit) applies the cell non-linearities, the first 64 threads write the row's 4 new hidden values as three
//     8-byte piece groups into the next step's planes.
// Measured (2 x 512, scripts/dev/bench_lstm_mid.py, launch to launch): 6.7 - 7.0 us per step at 9 .. 32 rows against 8.2
// (and 7.8 - 10.7 for lstm_fewrows_kernel at 9 .. 16), 8.8 at 36 and 9.9 at 64 against 10.1 and 11.0; shallow rings are as
// good as deep ones (profiles/r06i_lstm_mid16_ring_lab.txt); the evaluation pass of the configs[3] stand-in 40.0 -> 37.7 ms.
// Same arithmetic as everywhere (three bf16 pieces per operand, six products per fp32 product, fp32 accumulation); the
// sums are grouped differently from lstm_mid_x3_kernel (32 k per instruction), so the two agree to rounding, not to the bit.
// One workgroup per CU, one wave per SIMD, stores only in the finish (bf16x3.h).
#include "bf16x3.h"
#include "gemm_epilogue.h"

namespace empose {

namespace lh3 {
constexpr int BM = 64, BU = 4, NC = 16, NT = 256;
constexpr int PLD = BM + 4;                                   // row stride of a partial-sum column (floats)
constexpr int PART_FLOATS = 4 * NC * PLD;                     // [wave][column][row]
constexpr int HX_FLOATS = BM * (BU + 1);
constexpr size_t LDS_BYTES = 84 * 1024;                       // > half a CU: one workgroup per CU
static_assert((PART_FLOATS + HX_FLOATS) * 4 <= (int)LDS_BYTES, "LDS layout");
constexpr int FRAG = 512;                                     // bf16 elements of one fragment (1 KB)
}  // namespace lh3

typedef const __attribute__((address_space(1))) u32x4_t* lh3_gvec_t;
typedef const __attribute__((address_space(1))) unsigned short* lh3_gptr_t;
typedef float lh3_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned lh3_u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float lh3_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float lh3_tanh(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x)); }

template <int RT, int D>   // row tiles of 16 a workgroup multiplies (2: at most 32 rows, 4: at most 64); ring depth
__global__ __launch_bounds__(lh3::NT) void lstm_mid16_x3_kernel(LstmX3Args a) {
  X3_EXCLUSIVE_SIMD();
  using namespace lh3;
  extern __shared__ __attribute__((aligned(16))) float part[];
  float* hx = part + PART_FLOATS;
  const int H = a.H, B = a.B, F = a.F;
  const int jb = blockIdx.x, JB = H / BU, j0 = jb * BU;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const LstmX3Unit& U = a.unit[blockIdx.z];
  const int KS_h = H / 16, KS_in = U.ks_in;                   // k-steps of 16 of the planes
  const int K2_in = (KS_in + 1) / 2, K2_h = KS_h / 2, K2 = K2_in + K2_h;   // k-steps of 32 (the input's last may be half)
  const int t = U.t;
  const int RT32 = (B + 31) / 32;
  // the finishing thread's cell: row f_row, unit j0 + f_u
  const int f_row = tid & 63, f_u = tid >> 6;
  const int g_row = f_row, g_rowc = g_row < B ? g_row : B - 1;
  const int g_unit = j0 + f_u;
  const bool row_used = (RT == 4 || f_row < 32) && g_row < B;

  // ---- what the finish reads besides the sums, fetched now
  const int e_len = a.seq_lengths ? a.seq_lengths[g_rowc] : F;
  const size_t hc = (size_t)g_rowc * H + g_unit;
  const float e_c = U.c[hc], e_hp = U.h_prev[hc];
  float e_bias[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) e_bias[q] = U.bias[q * H + g_unit];

  lh3_f32x4 acc[RT][2];
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int h = 0; h < 2; ++h) acc[r][h] = lh3_f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- the wave's k-steps of 32: g = wave + 4 i
  u32x4_t fa[D][RT][3], fw[D][3];
  const int n_w = (K2 - wave + 3) / 4;
  const unsigned short* const p_in = U.a3_in; const unsigned short* const p_rec = U.a3_rec;
  const unsigned short* const p_wih = U.w3_ih; const unsigned short* const p_whh = U.w3_hh;
  // lane (row l15, k quarter lq): half lq & 1 of k-step 2 g + (lq >> 1) of the 32-row tile's plane
  const int lane_off = (l15 + 32 * (lq & 1)) * 8;
  auto load = [&, p_in, p_rec, p_wih, p_whh](u32x4_t (&A)[RT][3], u32x4_t (&W)[3], int i) {
    const int g = wave + 4 * i;
    const bool in = g < K2_in;
    const int k2 = in ? g : g - K2_in, ksn = in ? KS_in : KS_h;
    // the second k-step of 16 of this step; an odd input ends on a half step: its lanes read the first half again (finite
    // values against weights that are zero there)
    const int ks = 2 * k2 + ((2 * k2 + 1 < ksn) ? (lq >> 1) : 0);
    lh3_gptr_t ab = (lh3_gptr_t)(in ? p_in : p_rec) + ((size_t)ks * 3) * FRAG + lane_off;
    lh3_gptr_t wb = (lh3_gptr_t)(in ? p_wih : p_whh) + (((size_t)k2 * JB + jb) * 3) * FRAG + lane * 8;
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      // row tile of 16 r: 32-row tile r >> 1 (a tile past the batch reads the last one: rows never stored), rows (r & 1) * 16 ..
      const int rt32 = (r >> 1) < RT32 ? (r >> 1) : RT32 - 1;
      lh3_gptr_t ar = ab + (size_t)rt32 * ksn * 3 * FRAG + (r & 1) * 16 * 8;
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) A[r][pc] = *(lh3_gvec_t)(ar + pc * FRAG);
    }
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) W[pc] = *(lh3_gvec_t)(wb + pc * FRAG);
  };
  auto mma = [&](const u32x4_t (&A)[RT][3], const u32x4_t (&W)[3]) {
#pragma unroll
    for (int p = 0; p < 6; ++p)
#pragma unroll
      for (int r = 0; r < RT; ++r)
        acc[r][p & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, A[r][X3_PA[p]]),
                                                                __builtin_bit_cast(bf16x8_t, W[X3_PB[p]]), acc[r][p & 1], 0, 0, 0);
  };
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (d < n_w) load(fa[d], fw[d], d);
  for (int i0 = 0; i0 < n_w; i0 += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (i0 + d < n_w) mma(fa[d], fw[d]);                       // (uniform)
      if (i0 + d + D < n_w) load(fa[d], fw[d], i0 + d + D);
    }
  }

  // ---- partial sums -> LDS as [wave][column][row]: C/D of the 16x16 instruction: column = lane & 15, rows 4 (lane >> 4) ..
  {
    float* pw = part + (size_t)wave * NC * PLD;
#pragma unroll
    for (int r = 0; r < RT; ++r)
      *reinterpret_cast<lh3_f32x4*>(pw + l15 * PLD + r * 16 + 4 * lq) = acc[r][0] + acc[r][1];
  }
  __syncthreads();

  // ---- finish: thread (row, unit); column of gate q of unit u: q * 4 + u
  const bool live = t < e_len;
  float gsum[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float* ps = part + (q * BU + f_u) * PLD + f_row;
    gsum[q] = (RT == 4 || f_row < 32) ? ((ps[0] + ps[NC * PLD]) + ps[2 * NC * PLD]) + ps[3 * NC * PLD] : 0.f;
  }
  const float g_i = lh3_sigmoid(gsum[0] + e_bias[0]), g_f = lh3_sigmoid(gsum[1] + e_bias[1]);
  const float g_g = lh3_tanh(gsum[2] + e_bias[2]), g_o = lh3_sigmoid(gsum[3] + e_bias[3]);
  const float c_new = g_f * e_c + g_i * g_g;
  const float h_new = g_o * lh3_tanh(c_new);
  const float hv = live ? h_new : (a.seq_lengths ? e_hp : 0.f);
  if (row_used) {
    const size_t o = (size_t)g_row * H + g_unit;
    if (live) U.c[o] = c_new;
    U.h_next[o] = hv;
    if (U.y) U.y[((size_t)g_row * F + t) * U.y_ld + U.y_col + g_unit] = live ? h_new : 0.f;
  }
  hx[f_row * (BU + 1) + f_u] = hv;
  __syncthreads();
  // the row's 4 new hidden values as pieces: 8 bytes of the 16-byte group of units j0 & ~7 .. + 7 in the next step's planes
  if (tid < BM && tid < B && (RT == 4 || tid < 32)) {
    const float* src = hx + tid * (BU + 1);
    unsigned h[2], m[2], l[2];
    split_pair(src[0], src[1], h[0], m[0], l[0]);
    split_pair(src[2], src[3], h[1], m[1], l[1]);
    const int ks = j0 >> 4, ln = (tid & 31) + 32 * ((j0 & 15) >> 3);
    unsigned short* o = U.a3_out + ((((size_t)(tid >> 5)) * KS_h + ks) * 3) * FRAG + ln * 8 + (j0 & 7);
    *reinterpret_cast<lh3_u32x2*>(o) = lh3_u32x2{h[0], h[1]};
    *reinterpret_cast<lh3_u32x2*>(o + FRAG) = lh3_u32x2{m[0], m[1]};
    *reinterpret_cast<lh3_u32x2*>(o + 2 * FRAG) = lh3_u32x2{l[0], l[1]};
  }
}

#ifndef LH3_RING2
#define LH3_RING2 4
#endif
#ifndef LH3_RING4
#define LH3_RING4 3
#endif

bool lstm_mid16_shape_ok(int B, int H) { return B >= 1 && B <= lh3::BM && H % 32 == 0; }

hipError_t launch_lstm_mid16_x3(const LstmX3Args& a, hipStream_t stream) {
  if (a.n_units == 0) return hipSuccess;
  if (!lstm_mid16_shape_ok(a.B, a.H)) return hipErrorInvalidValue;
  dim3 grid(a.H / lh3::BU, 1, a.n_units);
  if (a.B <= 32) {
    auto* fn = lstm_mid16_x3_kernel<2, LH3_RING2>;
    if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(fn), lh3::LDS_BYTES)) return e;
    hipLaunchKernelGGL(fn, grid, dim3(lh3::NT), lh3::LDS_BYTES, stream, a);
  } else {
    auto* fn = lstm_mid16_x3_kernel<4, LH3_RING4>;
    if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(fn), lh3::LDS_BYTES)) return e;
    hipLaunchKernelGGL(fn, grid, dim3(lh3::NT), lh3::LDS_BYTES, stream, a);
  }
  return hipGetLastError();
}

}  // namespace empose
