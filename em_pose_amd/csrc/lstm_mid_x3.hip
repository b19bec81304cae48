// The LSTM wavefront step of MEDIUM batches (17 .. 256 rows: the batched evaluation driver runs chunk c of all recordings
// as one ragged batch -- scripts/evaluate_real.py, BASELINE configs[3]; reference nn/layers.py:133-157) on three bf16 pieces
// per operand (bf16x3.h), round 6.  Same arithmetic, same A planes ([32-row tile][k-step][piece], written by the producing
// step) and same argument block as lstm_x3.hip; what changes is the tile, because at 36 rows a launch is not bound by its
// matrix work but by how few workgroups share it and how long each one's dependent chain is:
//   * lstm_mid_kernel (fp32 MFMA, 32 rows x 16 units, 128 workgroups at 36 rows, operands staged through LDS tile by tile,
//     no pipelining) takes 17 us per step whatever the row count -- 55 % of the GPU time of the configs[3] stand-in
//     (profiles/r06_evaluate_real_kernel_stats.csv);
//   * here a workgroup owns up to 64 rows x 8 hidden units x 4 gates = ONE 32-column tile of one unit (the four gates of a
//     unit sit in the same tile: weights packed [k-step][8-unit block][piece], column = gate * 8 + unit; api.hip
//     pack_lstm_x3_mid), its four waves split K (wave w takes k-steps w, w + 4, ...: 16 steps of 6 or 12 MFMAs for K = 1024),
//     fragments straight from L2 into a three-deep register ring, no LDS and no barrier in the loop; the partial sums meet
//     in LDS (34 KB), then thread (row, 2 units) applies the cell non-linearities and the first 64 threads write the new
//     hidden row's three 16-byte piece groups where the next step's A fragments expect them.
//   128 workgroups (H = 512, two layers) of ~3 us instead of 128 of ~13.
// One workgroup per CU by its LDS request: the finish STORES while other workgroups' MFMAs would share the SIMD otherwise
// (scripts/dev/bf16_hazard_repro.md).  Two accumulators per row tile, products alternating between them: consecutive MFMAs
// never write the same accumulator.
#include "bf16x3.h"
#include "gemm_epilogue.h"

namespace empose {

namespace lm3 {
constexpr int BM = 64, BU = 8, NT = 256;
constexpr int PLD = BM + 4;                                   // row stride of a partial-sum column (floats)
constexpr int PART_FLOATS = 4 * 32 * PLD;                     // [wave][column][row]
constexpr int HX_FLOATS = BM * (BU + 1);                      // the new hidden values [row][unit] for the piece split
constexpr size_t LDS_BYTES = 84 * 1024;                       // > half a CU: one workgroup per CU
static_assert((PART_FLOATS + HX_FLOATS) * 4 <= (int)LDS_BYTES, "LDS layout");
constexpr int FRAG = 512;
constexpr int SG_MFMA = 0x008, SG_VMEM_RD = 0x020;
}  // namespace lm3

typedef const __attribute__((address_space(1))) u32x4_t* lm3_gvec_t;
typedef const __attribute__((address_space(1))) unsigned short* lm3_gptr_t;

__device__ __forceinline__ float lm3_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float lm3_tanh(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x)); }

template <int RTS, int D>   // row tiles of 32 a workgroup multiplies: 1 (launches of at most 32 rows) or 2; ring depth
__global__ __launch_bounds__(lm3::NT) void lstm_mid_x3_kernel(LstmX3Args a) {
  X3_EXCLUSIVE_SIMD();
  using namespace lm3;
  extern __shared__ __attribute__((aligned(16))) float part[];
  float* hx = part + PART_FLOATS;
  const int H = a.H, B = a.B, F = a.F;
#ifdef LM3_LAB_TWICE_WGS   // (dev: twice the workgroups, each with every other k-step of its waves -- results are wrong)
  const int JB = H / BU, jb = blockIdx.x % JB, j0 = jb * BU, lab_half = blockIdx.x / JB;
#else
  const int jb = blockIdx.x, JB = H / BU, j0 = jb * BU;
#endif
  const int m0 = blockIdx.y * BM, rt0 = blockIdx.y * 2, RT = (B + 31) / 32;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const LstmX3Unit& U = a.unit[blockIdx.z];
  const int KS_h = H / 16, KS_in = U.ks_in, KS = KS_in + KS_h;
  const int t = U.t;
  // the finishing thread's cells: row f_row, units j0 + 2 f_up, + 1
  const int f_row = tid & 63, f_up = tid >> 6;
  const int g_row = m0 + f_row, g_rowc = g_row < B ? g_row : B - 1;
  const int g_unit = j0 + 2 * f_up;

  // ---- what the finish reads besides the sums, fetched now
  const int e_len = a.seq_lengths ? a.seq_lengths[g_rowc] : F;
  float e_c[2], e_hp[2], e_bias[4][2];
  {
    const size_t hc = (size_t)g_rowc * H + g_unit;
    e_c[0] = U.c[hc]; e_c[1] = U.c[hc + 1];
    e_hp[0] = U.h_prev[hc]; e_hp[1] = U.h_prev[hc + 1];
#pragma unroll
    for (int q = 0; q < 4; ++q) { e_bias[q][0] = U.bias[q * H + g_unit]; e_bias[q][1] = U.bias[q * H + g_unit + 1]; }
  }

  f32x16 acc[RTS][2];
#pragma unroll
  for (int r = 0; r < RTS; ++r)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[r][h][v] = 0.f;

  // ---- the wave's k-steps: g = wave + 4 i
  u32x4_t fa[D][RTS][3], fw[D][3];
#ifdef LM3_LAB_TWICE_WGS
  const int n_w = ((KS - wave + 3) / 4 + 1 - lab_half) / 2;
#elif defined(LM3_LAB_NOK)      // (dev, scripts/dev/lstm_mid_lab.sh: one k-step per wave -- what a launch costs without its K loop)
  const int n_w = 1;
#else
  const int n_w = (KS - wave + 3) / 4;
#endif
  const unsigned short* const p_in = U.a3_in; const unsigned short* const p_rec = U.a3_rec;
  const unsigned short* const p_wih = U.w3_ih; const unsigned short* const p_whh = U.w3_hh;
  auto load = [&, p_in, p_rec, p_wih, p_whh](u32x4_t (&A)[RTS][3], u32x4_t (&W)[3], int i) {
#ifdef LM3_LAB_TWICE_WGS
    int g = wave + 4 * (i + (lab_half ? ((KS - wave + 3) / 4 + 1) / 2 : 0));   // the other half of the wave's k-steps
#else
    int g = wave + 4 * i;
#endif
    g = g < KS ? g : KS - 1;                      // (past the wave's last step: fetched, never multiplied)
    const bool in = g < KS_in;
    const int ks = in ? g : g - KS_in, ksn = in ? KS_in : KS_h;
    lm3_gptr_t ab = (lm3_gptr_t)(in ? p_in : p_rec) + (((size_t)rt0 * ksn + ks) * 3) * FRAG + lane * 8;
    lm3_gptr_t wb = (lm3_gptr_t)(in ? p_wih : p_whh) + (((size_t)ks * JB + jb) * 3) * FRAG + lane * 8;
    // (a second row tile that does not exist reads the first again; its rows are >= B and never stored)
    const size_t rt_stride = rt0 + 1 < RT ? (size_t)ksn * 3 * FRAG : 0;
#pragma unroll
    for (int r = 0; r < RTS; ++r)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) A[r][pc] = *(lm3_gvec_t)(ab + r * rt_stride + pc * FRAG);
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) W[pc] = *(lm3_gvec_t)(wb + pc * FRAG);
  };
  auto mma = [&](const u32x4_t (&A)[RTS][3], const u32x4_t (&W)[3]) {
#pragma unroll
    for (int p = 0; p < 6; ++p)
#pragma unroll
      for (int r = 0; r < RTS; ++r)
        acc[r][p & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, A[r][X3_PA[p]]),
                                                                __builtin_bit_cast(bf16x8_t, W[X3_PB[p]]), acc[r][p & 1], 0, 0, 0);
  };
  // A ring of D k-steps: the fragments of D steps are in flight before the first product.  A step's operands come from the
  // memory side of the L2 (every launch streams its weight block; the A planes were written by the launch before), ~2 us
  // away: with two steps in flight (rounds 6a-6g) the 16 steps of a wave were eight round trips, 7.4 us per launch at
  // 32 rows; the ring costs registers only (24 per step and row tile) and the wave has all 512.
#pragma unroll
  for (int d = 0; d < D; ++d) load(fa[d], fw[d], d);
  for (int i0 = 0; i0 < n_w; i0 += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (i0 + d < n_w) mma(fa[d], fw[d]);                       // (uniform)
      if (i0 + d + D < n_w) load(fa[d], fw[d], i0 + d + D);
    }
  }

  // ---- partial sums -> LDS as [wave][column][row] (16-byte pieces of four consecutive rows of a column)
  {
    float* pw = part + (size_t)wave * 32 * PLD;
#pragma unroll
    for (int r = 0; r < RTS; ++r)
#pragma unroll
      for (int v = 0; v < 4; ++v)
        *reinterpret_cast<f32x4*>(pw + l31 * PLD + r * 32 + 8 * v + 4 * lh) =
            f32x4{acc[r][0][4 * v] + acc[r][1][4 * v], acc[r][0][4 * v + 1] + acc[r][1][4 * v + 1],
                  acc[r][0][4 * v + 2] + acc[r][1][4 * v + 2], acc[r][0][4 * v + 3] + acc[r][1][4 * v + 3]};
  }
  __syncthreads();

  // ---- finish: thread (row, 2 units); column of gate q of unit u: q * 8 + u
  const bool row_used = RTS == 2 || f_row < 32;
  float hv[2];
  const bool live = t < e_len;
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    float gsum[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float* ps = part + (q * BU + 2 * f_up + e) * PLD + f_row;
      gsum[q] = row_used ? ((ps[0] + ps[32 * PLD]) + ps[2 * 32 * PLD]) + ps[3 * 32 * PLD] : 0.f;
    }
    const float g_i = lm3_sigmoid(gsum[0] + e_bias[0][e]), g_f = lm3_sigmoid(gsum[1] + e_bias[1][e]);
    const float g_g = lm3_tanh(gsum[2] + e_bias[2][e]), g_o = lm3_sigmoid(gsum[3] + e_bias[3][e]);
    const float c_new = g_f * e_c[e] + g_i * g_g;
    const float h_new = g_o * lm3_tanh(c_new);
    hv[e] = live ? h_new : (a.seq_lengths ? e_hp[e] : 0.f);
    if (g_row < B && row_used) {
      const size_t hc = (size_t)g_row * H + g_unit + e;
      if (live) U.c[hc] = c_new;
      U.h_next[hc] = hv[e];
      if (U.y) U.y[((size_t)g_row * F + t) * U.y_ld + U.y_col + g_unit + e] = live ? h_new : 0.f;
    }
    hx[f_row * (BU + 1) + 2 * f_up + e] = hv[e];
  }
  __syncthreads();
  // the new hidden values as pieces, where the next step's (and the layer above's) A fragments expect them
  if (tid < BM && g_row < B && row_used) {
    const float* src = hx + tid * (BU + 1);
    const Pieces q = split8(src[0], src[1], src[2], src[3], src[4], src[5], src[6], src[7]);
    const int ks = j0 >> 4, ln = (g_row & 31) + 32 * ((j0 & 15) >> 3);
    unsigned short* o = U.a3_out + ((((size_t)(g_row >> 5)) * KS_h + ks) * 3) * FRAG + ln * 8;
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) *reinterpret_cast<u32x4_t*>(o + pc * FRAG) = q.p[pc];
  }
}

#ifndef LM3_RING1
#define LM3_RING1 8
#endif
#ifndef LM3_RING2
#define LM3_RING2 6
#endif

hipError_t launch_lstm_mid_x3(const LstmX3Args& a, hipStream_t stream) {
  if (a.n_units == 0) return hipSuccess;
#ifdef LM3_LAB_TWICE_WGS
  dim3 grid(2 * a.H / lm3::BU, (a.B + lm3::BM - 1) / lm3::BM, a.n_units);
#else
  dim3 grid(a.H / lm3::BU, (a.B + lm3::BM - 1) / lm3::BM, a.n_units);
#endif
  if (a.B <= 32) {
    auto* fn = lstm_mid_x3_kernel<1, LM3_RING1>;
    if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(fn), lm3::LDS_BYTES)) return e;
    hipLaunchKernelGGL(fn, grid, dim3(lm3::NT), lm3::LDS_BYTES, stream, a);
  } else {
    auto* fn = lstm_mid_x3_kernel<2, LM3_RING2>;
    if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(fn), lm3::LDS_BYTES)) return e;
    hipLaunchKernelGGL(fn, grid, dim3(lm3::NT), lm3::LDS_BYTES, stream, a);
  }
  return hipGetLastError();
}

}  // namespace empose
