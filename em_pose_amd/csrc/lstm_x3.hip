// The LSTM wavefront step of large batches (lstm.hip, lstm_chain_kernel: reference nn/layers.py:133-157) with every fp32
// product formed on the bf16 matrix path from three bf16 pieces per operand (bf16x3.h): fp32-equivalent arithmetic, six
// v_mfma_f32_32x32x16_bf16 per 16 k instead of 64 v_mfma_f32_16x16x4_f32-quarters.
//
// What changes against the fp32 kernel is where the operands come from.  At the bf16 rate a 64-wide K tile is 1.5 k cycles
// of matrix work, and staging both operands through LDS (global -> registers -> LDS -> fragments) no longer hides under
// it.  So nothing is staged:
//   * W_ih / W_hh are split ONCE when the model is created and stored in fragment order, the four gates of a 32-unit block
//     side by side: [k-step][unit block][gate][piece] -> one 1 KB wave fragment (api.hip pack_lstm_x3);
//   * the hidden states are split by the workgroup that PRODUCES them: besides h (fp32, for the state hand-over and rows
//     past their length) a step writes h's three pieces in A-fragment order [32-row tile][k-step][piece] -> 1 KB, so that
//     the next step (and the layer above) loads fragments, not rows;
//   * the stored input sequence is split once per forward by lstm_split_rows_kernel.
// A workgroup owns 64 batch rows x 32 hidden units x 4 gates (8 accumulator tiles of 32 x 32) and walks the chain of its
// units (layer 0: K = input + H, then layer 1: K = 2 H).  Its four waves SPLIT K: wave w takes the k-steps w, w + 4, ...
// of the unit's whole K, every wave keeps the full 64 x 128 tile in registers (128 VGPRs), 18 fragment loads per 48 MFMAs,
// no LDS and no barrier in the K loop.  The partial gate sums meet in LDS (4 x 34 KB) and are added in wave order
// (deterministic); then thread (row, group of 8 units) applies the cell non-linearities and stores c, h, y and the three
// 16-byte piece groups of its 8 new hidden values.
// One wave per SIMD (139 KB of LDS per workgroup), like everything in this tree that issues the bf16 32x32x16 MFMA.
#include "bf16x3.h"
#include "gemm_epilogue.h"

namespace empose {

namespace lx {
constexpr int BM = 64, BU = 32, NT = 256;
constexpr int PLD = BM + 4;                                          // row stride of a partial-sum column (floats)
constexpr size_t LDS_BYTES = (size_t)4 * 4 * BU * PLD * sizeof(float);   // [wave][gate][unit][row]: 139,264 bytes
constexpr int FRAG = 512;                                            // bf16 elements of one wave fragment (1 KB)
constexpr int SG_MFMA = 0x008, SG_VMEM_RD = 0x020;
}  // namespace lx

typedef const __attribute__((address_space(1))) u32x4_t* lx_gvec_t;
typedef const __attribute__((address_space(1))) unsigned short* lx_gptr_t;

#define LX_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)

__device__ __forceinline__ float lx_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float lx_tanh(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x)); }

// Row-major fp32 matrices -> A-fragment-order pieces.  Matrix z: rows src + z * z_stride + row * row_stride, K columns
// (K % 4 == 0); output z: dst + z * dst_z_stride as [32-row tile][k-step of 16][piece][lane][8] with
// lane = (row & 31) + 32 * ((k & 15) >> 3).  Rows >= B and columns >= K of the padded block are zero.
__global__ __launch_bounds__(256) void lstm_split_rows_kernel(const float* __restrict__ src, long row_stride, long z_stride,
                                                              int B, int K, int KS, unsigned short* __restrict__ dst,
                                                              long dst_z_stride) {
  const int RT = (B + 31) / 32, KG = KS * 2;
  const long per_z = (long)RT * 32 * KG;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= per_z) return;
  const int z = blockIdx.y;
  const int row = (int)(i / KG), kg = (int)(i % KG);
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = 0.f;
  if (row < B) {
    const float* p = src + (long)z * z_stride + (long)row * row_stride + kg * 8;
    if (kg * 8 < K) { const f32x4 q = *reinterpret_cast<const f32x4*>(p); v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3]; }
    if (kg * 8 + 4 < K) { const f32x4 q = *reinterpret_cast<const f32x4*>(p + 4); v[4] = q[0]; v[5] = q[1]; v[6] = q[2]; v[7] = q[3]; }
  }
  const Pieces q = split8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
  const int rt = row >> 5, ks = kg >> 1, lane = (row & 31) + 32 * (kg & 1);
  unsigned short* o = dst + (long)z * dst_z_stride + (((long)rt * KS + ks) * 3) * lx::FRAG + lane * 8;
#pragma unroll
  for (int pc = 0; pc < 3; ++pc) *reinterpret_cast<u32x4_t*>(o + pc * lx::FRAG) = q.p[pc];
}

#ifdef LX_LAB_TIMES      // (scripts/dev/lstm_x3_lab.hip: shader-clock stamps of thread 0 of every workgroup)
__device__ long long* lx_lab_times;
#define LX_STAMP(i) do { if (threadIdx.x == 0) lx_lab_times[(blockIdx.x + gridDim.x * blockIdx.y) * 16 + (i)] = clock64(); } while (0)
#else
#define LX_STAMP(i) do { } while (0)
#endif

__global__ __launch_bounds__(lx::NT) void lstm_chain_x3_kernel(LstmX3Args a) {
  X3_EXCLUSIVE_SIMD();
  using namespace lx;
  LX_STAMP(0);
  extern __shared__ __attribute__((aligned(16))) float part[];
  const int H = a.H, B = a.B, F = a.F;
  const int jb = blockIdx.x, JB = H / BU, j0 = jb * BU;
  const int m0 = blockIdx.y * BM, rt0 = blockIdx.y * 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int KS_h = H / 16;
  // the finishing thread's element group: row f_row, units j0 + 8 f_ug .. + 7 (all four gates).  (Four lanes per row --
  // whole 128-byte lines of c / h / y per row -- measured the same: the finish is its 128 LDS reads and 40 transcendental
  // functions per thread, 5.1 k of the 39 k clocks a unit takes, scripts/dev/lstm_x3_lab.hip with -DLX_LAB_TIMES)
  const int f_row = tid & 63, f_ug = tid >> 6;
  const int g_row = m0 + f_row, g_rowc = g_row < B ? g_row : B - 1;
  const int g_unit = j0 + f_ug * 8;

  const int u_beg = blockIdx.z * a.units_per_block;
  const int u_end = u_beg + a.units_per_block < a.n_units ? u_beg + a.units_per_block : a.n_units;
  for (int u = u_beg; u < u_end; ++u) {
    const LstmX3Unit& U = a.unit[u];
    const int KS_in = U.ks_in, KS = KS_in + KS_h;
    const int t = U.t;
    // ---- what the finish reads besides the sums, fetched now (latency under the K loop)
    f32x4 e_c[2], e_hp[2];
    float e_bias[4][8];
    const int e_len = a.seq_lengths ? a.seq_lengths[g_rowc] : F;
    {
      const size_t hc = (size_t)g_rowc * H + g_unit;
      e_c[0] = *reinterpret_cast<const f32x4*>(U.c + hc);
      e_c[1] = *reinterpret_cast<const f32x4*>(U.c + hc + 4);
      e_hp[0] = *reinterpret_cast<const f32x4*>(U.h_prev + hc);
      e_hp[1] = *reinterpret_cast<const f32x4*>(U.h_prev + hc + 4);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(U.bias + q * H + g_unit);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(U.bias + q * H + g_unit + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { e_bias[q][e] = b0[e]; e_bias[q][4 + e] = b1[e]; }
      }
    }

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][q][r] = 0.f;

    // ---- the wave's k-steps: g = wave + 4 i.  Fragments of step g: A from the input planes (g < KS_in) or the recurrent
    // ones, W from the matching matrix; everything is a wave-uniform base plus lane * 16 bytes.
    u32x4_t fa[3][2][3], fw[3][4][3];
    const int n_w = (KS - wave + 3) / 4;
    // (the unit's operand pointers in scalar registers for the whole K loop: read from the kernel arguments inside it they
    // cost a scalar load and a wait -- a bubble in the matrix pipe -- per fragment set)
    const unsigned short* const p_in = U.a3_in; const unsigned short* const p_rec = U.a3_rec;
    const unsigned short* const p_wih = U.w3_ih; const unsigned short* const p_whh = U.w3_hh;
    auto load = [&, p_in, p_rec, p_wih, p_whh](u32x4_t (&A)[2][3], u32x4_t (&W)[4][3], int i) {
      int g = wave + 4 * i;
      g = g < KS ? g : KS - 1;                      // (past the wave's last step: fetched, never multiplied)
      const bool in = g < KS_in;
      const int ks = in ? g : g - KS_in, ksn = in ? KS_in : KS_h;
      lx_gptr_t ab = (lx_gptr_t)(in ? p_in : p_rec) + (((size_t)rt0 * ksn + ks) * 3) * FRAG + lane * 8;
      lx_gptr_t wb = (lx_gptr_t)(in ? p_wih : p_whh) + ((((size_t)ks * JB + jb) * 4) * 3) * FRAG + lane * 8;
      // (an odd number of 32-row tiles: the last workgroup's second tile does not exist in the piece planes -- it reads its
      // first tile again instead of one tile past the plane; those rows are >= B and never stored)
      const size_t rt_stride = rt0 + 1 < (B + 31) / 32 ? (size_t)ksn * 3 * FRAG : 0;
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) A[r][pc] = *(lx_gvec_t)(ab + r * rt_stride + pc * FRAG);
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) W[q][pc] = *(lx_gvec_t)(wb + (q * 3 + pc) * FRAG);
    };
    auto mma = [&](const u32x4_t (&A)[2][3], const u32x4_t (&W)[4][3]) {
#pragma unroll
      for (int p = 0; p < 6; ++p)
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            acc[r][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, A[r][X3_PA[p]]),
                                                                __builtin_bit_cast(bf16x8_t, W[q][X3_PB[p]]), acc[r][q], 0, 0, 0);
    };
    auto pattern = [&]() {   // 48 MFMAs with the 18 fragment loads of the step after next between them
#pragma unroll
      for (int q = 0; q < 18; ++q) { LX_SGB(SG_MFMA, 2); LX_SGB(SG_VMEM_RD, 1); }
      LX_SGB(SG_MFMA, 12);
    };
    load(fa[0], fw[0], 0);
    load(fa[1], fw[1], 1);
    int i = 0;
#ifdef LX_LAB_NOLOAD      // (lab ablation, scripts/dev/lstm_x3_lab.hip: the K loop on the first fragments only)
    load(fa[2], fw[2], 2);
    for (; i + 3 <= n_w; i += 3) { mma(fa[0], fw[0]); mma(fa[1], fw[1]); mma(fa[2], fw[2]); }
#else
    for (; i + 3 <= n_w; i += 3) {
      load(fa[2], fw[2], i + 2);
      mma(fa[0], fw[0]);
      pattern();
      load(fa[0], fw[0], i + 3);
      mma(fa[1], fw[1]);
      pattern();
      load(fa[1], fw[1], i + 4);
      mma(fa[2], fw[2]);
      pattern();
    }
#endif
    if (i < n_w) mma(fa[0], fw[0]);
    if (i + 1 < n_w) mma(fa[1], fw[1]);
    LX_STAMP(1 + 5 * (u - u_beg));

    // ---- partial sums -> LDS as [wave][gate][unit][row]: 16-byte pieces of four consecutive rows (the C/D layout has rows
    // 8 q + 4 lh .. + 3 of a column in one lane); the row stride of 68 floats spreads the 32 units of a store over the banks
#ifdef LX_LAB_NOPART     // (lab ablation: no exchange, no finish -- the sums leave through one predicated store)
    {
      float sacc = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int v = 0; v < 16; ++v) sacc += acc[r][q][v];
      if (sacc == 123.456f) U.c[tid] = sacc + e_c[0][0] + e_hp[0][0] + e_bias[0][0];
      continue;
    }
#endif
    if (u > u_beg) __syncthreads();   // the previous unit's finish has read its sums
    {
      float* pw = part + (size_t)wave * 4 * BU * PLD;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int v = 0; v < 4; ++v)
            *reinterpret_cast<f32x4*>(pw + (q * BU + l31) * PLD + r * 32 + 8 * v + 4 * lh) =
                f32x4{acc[r][q][4 * v], acc[r][q][4 * v + 1], acc[r][q][4 * v + 2], acc[r][q][4 * v + 3]};
    }
    LX_STAMP(2 + 5 * (u - u_beg));
    __syncthreads();
    LX_STAMP(3 + 5 * (u - u_beg));

#ifdef LX_LAB_NOFINISH   // (lab ablation: the exchange, but no cell arithmetic and no stores)
    if (part[tid] == 123.456f) U.c[tid] = part[tid + 1] + e_c[0][0] + e_hp[0][0] + e_bias[0][0];
    continue;
#endif
    // ---- finish: thread (row, 8 units)
    float hv[8], cv[8], yv[8];
    const bool live = t < e_len;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float gsum[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float* ps = part + (q * BU + f_ug * 8 + e) * PLD + f_row;
        gsum[q] = ((ps[0] + ps[4 * BU * PLD]) + ps[2 * 4 * BU * PLD]) + ps[3 * 4 * BU * PLD];
      }
      const float g_i = lx_sigmoid(gsum[0] + e_bias[0][e]), g_f = lx_sigmoid(gsum[1] + e_bias[1][e]);
      const float g_g = lx_tanh(gsum[2] + e_bias[2][e]), g_o = lx_sigmoid(gsum[3] + e_bias[3][e]);
      const float c_old = e_c[e >> 2][e & 3], h_old = e_hp[e >> 2][e & 3];
      const float c_new = g_f * c_old + g_i * g_g;
      const float h_new = g_o * lx_tanh(c_new);
      cv[e] = live ? c_new : c_old;
      hv[e] = live ? h_new : (a.seq_lengths ? h_old : 0.f);
      yv[e] = live ? h_new : 0.f;
    }
    LX_STAMP(4 + 5 * (u - u_beg));
    if (g_row < B) {
      const size_t hc = (size_t)g_row * H + g_unit;
      *reinterpret_cast<f32x4*>(U.c + hc) = f32x4{cv[0], cv[1], cv[2], cv[3]};
      *reinterpret_cast<f32x4*>(U.c + hc + 4) = f32x4{cv[4], cv[5], cv[6], cv[7]};
      *reinterpret_cast<f32x4*>(U.h_next + hc) = f32x4{hv[0], hv[1], hv[2], hv[3]};
      *reinterpret_cast<f32x4*>(U.h_next + hc + 4) = f32x4{hv[4], hv[5], hv[6], hv[7]};
      if (U.y) {
        float* yo = U.y + ((size_t)g_row * F + t) * U.y_ld + U.y_col + g_unit;
        *reinterpret_cast<f32x4*>(yo) = f32x4{yv[0], yv[1], yv[2], yv[3]};
        *reinterpret_cast<f32x4*>(yo + 4) = f32x4{yv[4], yv[5], yv[6], yv[7]};
      }
      // the new hidden values as pieces, where the next step's (and the layer above's) A fragments expect them
      const Pieces q = split8(hv[0], hv[1], hv[2], hv[3], hv[4], hv[5], hv[6], hv[7]);
      const int ks = g_unit >> 4, ln = (g_row & 31) + 32 * ((g_unit & 15) >> 3);
      unsigned short* o = U.a3_out + ((((size_t)(g_row >> 5)) * KS_h + ks) * 3) * FRAG + ln * 8;
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) *reinterpret_cast<u32x4_t*>(o + pc * FRAG) = q.p[pc];
    }
    LX_STAMP(5 + 5 * (u - u_beg));
  }
}

hipError_t launch_lstm_split_rows(const float* src, long row_stride, long z_stride, int n_z, int B, int K, int KS,
                                  unsigned short* dst, long dst_z_stride, hipStream_t stream) {
  const long per_z = (long)((B + 31) / 32) * 32 * KS * 2;
  dim3 grid((unsigned)((per_z + 255) / 256), n_z);
  hipLaunchKernelGGL(lstm_split_rows_kernel, grid, dim3(256), 0, stream, src, row_stride, z_stride, B, K, KS, dst,
                     dst_z_stride);
  return hipGetLastError();
}

hipError_t launch_lstm_chain_x3(const LstmX3Args& a, hipStream_t stream) {
  if (a.n_units == 0) return hipSuccess;
  if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(lstm_chain_x3_kernel), lx::LDS_BYTES)) return e;
  dim3 grid(a.H / lx::BU, (a.B + lx::BM - 1) / lx::BM, (a.n_units + a.units_per_block - 1) / a.units_per_block);
  hipLaunchKernelGGL(lstm_chain_x3_kernel, grid, dim3(lx::NT), lx::LDS_BYTES, stream, a);
  return hipGetLastError();
}

}  // namespace empose
