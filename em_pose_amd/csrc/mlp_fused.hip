// A whole update MLP (reference nn/layers.py:46-77: Linear-BN-PReLU, hidden blocks, Linear; eval mode) in ONE launch,
// with the activations resident in LDS from the first layer to the last.
//
// A workgroup owns 64 batch rows of one net.  Its activations [64][<= 512] live in ONE LDS buffer (rows padded to 516
// floats = conflict-free ds_read_b128 for the 32x32x2 operand layout, 132 KB of the CU's 160 KB): a layer reads its A
// operand straight out of that buffer for the whole K loop and, once every wave has finished the loop, its epilogue
// (bias, folded BatchNorm, PReLU) overwrites the buffer in place with the next layer's input.  Between the network
// input x and the 66- / 10-column outputs nothing is stored to or loaded from global memory except the weights:
//   * weights never touch LDS: they are packed once (api.hip pack_fragments) in MFMA fragment order, so a wave's B
//     fragment of a k-group of 8 is one coalesced 1 KB global load straight into registers, prefetched three k-groups
//     ahead in a four-slot register ring (the weights of both nets are L2-resident);
//   * no barrier inside a layer's K loop (its A operand is static): two workgroup barriers per layer in total.
// The four waves split a layer's output columns into 128-column quarters (wave tile 64 x 128 = 2 x 4 MFMA tiles; eight
// waves of 64 x 64 measured 1 % slower); the K loop runs four k-groups per iteration with the interleaving pinned by
// `sched_group_barrier`:
//   k-group g : 32 MFMAs | A fragments of group g+1 (LDS) | B ring slot (g+3) % 4 <- group g+3 (global)
// Narrow layers (N <= 128, the output layers) split rows AND columns over the waves (32 x 64 each) so that a 66- or
// 10-column layer does not cost a 512-column one, and write to the net's output instead of the LDS buffer.
// Skip connections would need the block input kept besides the activations: such nets take the layer-by-layer path.
#include "bf16x3.h"
#include "gemm_epilogue.h"
#include "feat_rows.h"

namespace empose {

namespace fm {
constexpr int BM = 64, NT = 256;
constexpr int LDA = FUSED_MAX_WIDTH + 4;                     // activation row stride in LDS (floats)
constexpr size_t LDS_BYTES = (size_t)BM * LDA * sizeof(float) + 64;   // + the look-ahead of the last fragment read
constexpr int SG_MFMA = 0x008, SG_VMEM_RD = 0x020, SG_DS_RD = 0x100;
}  // namespace fm

// LDS layout of the single-layer row-block kernels below -- ONE definition: the kernels take their strides and region
// offsets from it and the launchers their byte count (a launcher that re-derived the size by hand once sized
// heads_rows_kernel for its staged rows only, while the transposed result that later overwrites them was larger).
//   [0, a)        the staged A block: row-major [rows][lda] plus the 16 floats the last fragment's look-ahead reads
//                 (fused_layer_t's fread(g + 4)), or K-major [kpad][64] (look-ahead clamped, nothing behind it)
//   [0, ct)       C^T [32-column tiles x 32][64] written over the A block after the K loop (OUT_T == 1), if any
//   [extra_off, extra_off + extra)   a kernel-specific region: behind both blocks, or -- `extra_over_a`: it is only used
//                 once the A block is dead -- behind C^T alone
struct RowsLds {
  int kpad, lda, extra_off, total;   // floats
  __host__ __device__ constexpr size_t bytes() const { return (size_t)total * sizeof(float); }
};
__host__ __device__ constexpr RowsLds rows_lds(int K, int rows, bool a_kmajor, int ct_cols = 0, int extra = 0,
                                               bool extra_over_a = false) {
  const int kpad = (((K + 7) / 8 + 3) & ~3) * 8;   // whole quads of k-groups: a multiple of 32
  const int lda = a_kmajor ? 64 : kpad + 4;
  const int a = a_kmajor ? kpad * 64 : rows * lda + 16;
  const int ct = ((ct_cols + 31) / 32) * 32 * 64;
  const int extra_off = extra_over_a ? ct : (a > ct ? a : ct);
  const int end = extra_off + extra;
  return RowsLds{kpad, lda, extra_off, end > a ? (end > ct ? end : ct) : (a > ct ? a : ct)};
}

#define FM_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)

#ifdef EMPOSE_FUSED_TRACE   // dev lab only: shader-clock stamps of block (0,0): per layer start, loop start, loop end, end
__device__ long long g_fused_trace[64];
#define FM_STAMP(i) if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) g_fused_trace[(i)] = clock64();
#else
#define FM_STAMP(i)
#endif

typedef const __attribute__((address_space(1))) f32x4* fm_gvec_t;
typedef const __attribute__((address_space(1))) char* fm_gbyte_t;

// One layer for the workgroup's rows; the input activations are in `act` (row stride lda, columns [0, 8 * KG4) valid or
// zero).  The wave computes WM x WN tiles of 32 x 32 starting at (row_tile0, col_tile0).
// OUT_T (last layer only): 0 the net's output rows | 1 C^T to LDS as [column][64], rotated (callers that go on
// working on it) | 2 straight to the tile-layout output [tile][column][64] as 16-byte pieces of four consecutive rows
// (the C/D layout has rows 4q .. 4q+3 of a column in one lane), no LDS and no barrier after the K loop.
// A_KMAJOR: the A operand lies in LDS as [k][64 rows] (a tile-layout block copied as it is: no transposition on the way
// in, conflict-free 4-byte reads) instead of [64 rows][lda].
template <int WM, int WN, int OUT_T = 0, bool A_KMAJOR = false>
__device__ __forceinline__ void fused_layer_t(const FusedNet& net, const FusedLayer& L, int M, int m0, float* act,
                                              int lda, int row_tile0, int col_tile0, int layer_index, bool last) {
  using namespace fm;
  FM_STAMP(4 * layer_index)
  constexpr int NMMA = WM * WN * 4;           // MFMAs per k-group of 8
  const int lane = threadIdx.x & 63;
  const int l31 = lane & 31, lh = lane >> 5;
  const int K = L.K, N = L.N;
  const int NT32 = (N + 31) / 32;             // column tiles of the packed weights
  const int KG4 = ((K + 7) / 8 + 3) & ~3;     // k-groups of the packed weights (padded with zeros to a multiple of 4)
  if (col_tile0 >= NT32) {   // wave-uniform: nothing of this layer falls to this wave; keep the two barriers
    if (OUT_T != 2) {
      __syncthreads();
      __syncthreads();
    }
    return;
  }
  // Column tiles past the layer's width are clamped to the last one: fetched and multiplied like the others (no
  // branch in the pipelined loop), never stored.
  // B side: fragment (kg, nt) starts at byte ((kg * NT32 + nt) * 64) * 16 and this lane owns 16 bytes of it: a scalar
  // base per k-group plus a constant 32-bit lane offset per column tile (no vector address arithmetic in the loop).
  unsigned b_voff[WN];
#pragma unroll
  for (int j = 0; j < WN; ++j)
    b_voff[j] = (unsigned)((col_tile0 + j < NT32 ? col_tile0 + j : NT32 - 1) * 1024 + lane * 16);
  fm_gbyte_t wb = (fm_gbyte_t)L.W;
  // A fragments: row tile i adds 32 rows, group g adds 8 columns (K-major: 8 rows of 64)
  const float* a_rd = A_KMAJOR ? act + (lh * 4) * 64 + row_tile0 * 32 + l31 : act + (row_tile0 * 32 + l31) * lda + lh * 4;

  // epilogue constants of this lane's columns, fetched now so that their latency hides under the K loop.  Columns past
  // N (zero padding of the next layer's K) get scale = shift = 0, so the hidden epilogue needs no select for them.
  // PReLU with a slope in [0, 1] is max(y, slope * y): the same bits as the select of the layer-by-layer epilogue, one
  // instruction fewer; other slopes take the select.
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
  f32x4 fa[2][WM];              // A fragments, double-buffered over the k-groups
  f32x4 fb[4][WN];              // B fragment ring: slot s holds k-group 4 t + s

  auto fread = [&](int g, f32x4 (&a)[WM]) {
    if (A_KMAJOR) {
      const int gc = g < KG4 ? g : KG4 - 1;   // (the look-ahead past the last group stays inside the block)
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) a[i][e] = a_rd[(gc * 8 + e) * 64 + i * 32];
    } else {
#pragma unroll
      for (int i = 0; i < WM; ++i) a[i] = *reinterpret_cast<const f32x4*>(a_rd + i * 32 * lda + g * 8);
    }
  };
  auto bload = [&](f32x4 (&b)[WN], int kg) {   // k-group kg of the packed weights (clamped: fetched, never used)
    const int kc = kg < KG4 ? kg : KG4 - 1;
    fm_gbyte_t p = wb + (size_t)kc * NT32 * 1024;
#pragma unroll
    for (int j = 0; j < WN; ++j) b[j] = *(fm_gvec_t)(p + b_voff[j]);
  };
  auto mma = [&](const f32x4 (&a)[WM], const f32x4 (&b)[WN]) {   // consecutive MFMAs go to different accumulator tiles
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
  };
  auto pattern = [&]() {   // one k-group: WM fragment reads and WN weight loads spread over the NMMA MFMAs
    constexpr int NDS = A_KMAJOR ? 4 * WM : WM;   // LDS reads per k-group
    constexpr int step = NMMA >= 2 * (NDS + WN) + 2 ? 2 : 1;
    static_assert(NMMA >= step * (NDS + WN), "k-group too small for its memory operations");
#pragma unroll
    for (int q = 0; q < NDS; ++q) { FM_SGB(SG_MFMA, step); FM_SGB(SG_DS_RD, 1); }
#pragma unroll
    for (int q = 0; q < WN; ++q) { FM_SGB(SG_MFMA, step); FM_SGB(SG_VMEM_RD, 1); }
    FM_SGB(SG_MFMA, NMMA - step * (NDS + WN));
  };

#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- prologue: weight ring slots 0..2, A fragments of group 0
  bload(fb[0], 0);
  bload(fb[1], 1);
  bload(fb[2], 2);
  fread(0, fa[0]);
  FM_STAMP(4 * layer_index + 1)

  auto quad = [&](int g) {   // four k-groups: ring slots 0..3 in turn
    fread(g + 1, fa[1]);
    bload(fb[3], g + 3);
    mma(fa[0], fb[0]);
    pattern();
    fread(g + 2, fa[0]);
    bload(fb[0], g + 4);
    mma(fa[1], fb[1]);
    pattern();
    fread(g + 3, fa[1]);
    bload(fb[1], g + 5);
    mma(fa[0], fb[2]);
    pattern();
    fread(g + 4, fa[0]);   // past the last group: columns inside the padded row, never used
    bload(fb[2], g + 6);
    mma(fa[1], fb[3]);
    pattern();
  };
  // The first iteration is peeled: the loop header then merges two states with the same outstanding loads, and the
  // compiler's wait-count insertion keeps the ring's prefetch distance (vmcnt(10)) instead of draining every load at
  // the top of each iteration (vmcnt(0)).  Measured floor of this loop: 8272 cycles per four k-groups with no memory
  // operations (8192 ideal), 8700 with the sixteen 1 KB weight-fragment loads, i.e. ~23 cycles per load.
  // Whole quads of real k-groups, then the 1..3 groups that are left (K = 200: 25 groups, K = 296: 37) from what the last
  // quad has already fetched -- its look-ahead holds the A fragment of group KGQ and the weights of KGQ .. KGQ + 2 -- instead
  // of a fourth quad that multiplies the zero padding (3 of 28 / 40 groups).  Same products in the same k order.
  const int KG = (K + 7) / 8;
  const int KGQ = KG >= 4 ? (KG & ~3) : KG4, tail = KG >= 4 ? KG - KGQ : 0;   // (fewer than four groups: one padded quad)
  quad(0);
  for (int g = 4; g < KGQ; g += 4) quad(g);
  if (tail >= 1) mma(fa[0], fb[0]);
  if (tail >= 2) {
    fread(KGQ + 1, fa[1]);
    mma(fa[1], fb[1]);
  }
  if (tail >= 3) {
    fread(KGQ + 2, fa[0]);
    mma(fa[0], fb[2]);
  }
  FM_STAMP(4 * layer_index + 2)

  if (last && OUT_T == 2) {
    float* dst = net.out + (size_t)(m0 >> 6) * net.ld_out * 64;
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int n = (col_tile0 + j) * 32 + l31;
      if (col_tile0 + j >= NT32 || n >= net.ld_out) continue;
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = (row_tile0 + i) * 32 + 8 * q + 4 * lh;
          *reinterpret_cast<f32x4*>(dst + (size_t)n * 64 + row) =
              f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
        }
    }
    return;
  }
  __syncthreads();   // every wave has read its last A fragment: the buffer may be overwritten
  if (last && OUT_T == 1) {
    // C^T into the (now free) activation buffer as [column][64 rows], rows rotated by the column so that the 32 lanes of
    // a store (32 columns, one row) hit 32 banks; gemm_rows_t_kernel copies it out as whole 256-byte columns.
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      if (col_tile0 + j >= NT32) continue;
      const int n = (col_tile0 + j) * 32 + l31;
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (row_tile0 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          act[n * 64 + ((row + n) & 63)] = acc[i][j][r];
        }
    }
  } else if (last) {
    // last layer: to the net's output
    GemmProb p;
    p.C = net.out; p.ldc = net.ld_out;
    p.M = M; p.N = N; p.K = K;
    p.scale = L.scale; p.shift = L.shift; p.resid = nullptr; p.ldr = 0;
    p.act = L.act; p.slope = L.slope;
    p.A = nullptr; p.W = nullptr; p.lda = 0; p.ldw = 0;
    if (col_tile0 < NT32) epilogue<WM, WN>(p, acc, m0 + row_tile0 * 32, col_tile0 * 32, l31, lh);
  } else {
    // hidden layer: scale/shift (bias, folded BatchNorm) and PReLU, in place into the activation buffer.
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int n = (col_tile0 + j) * 32 + l31;
      if (col_tile0 + j >= NT32) continue;            // wave-uniform; NT32 * 32 == the next layer's padded K
      // columns past N: 0 * acc + 0 = the next layer's zero padding
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
  __syncthreads();   // the next layer's A operand is complete
  FM_STAMP(4 * layer_index + 3)
}

// ---------------------------------------------------------------------------------------------------------------------
// The single-layer row-block product on the bf16 matrix path, every fp32 product from three bf16 pieces per operand
// (bf16x3.h; round 5).  Same operands and output modes as fused_layer_t<WM, WN, OUT_T, A_KMAJOR> with `last` set -- the A
// block in LDS as fp32 (row-major or K-major), split in registers as it is read; the weights as three bf16 pieces in
// fragment order (api.hip pack_fragments_x3_raw: [k-step of 16][32-column tile][piece] -> 1 KB) --, the K loop of
// mlp_fused_x3.hip's x3_layer (k-steps as fenced chunks, weights ahead in a register ring), plus plain steps for what is
// left behind the whole periods of the pipeline (K = 200: 12 pipelined steps + 1).
// ---------------------------------------------------------------------------------------------------------------------
typedef const __attribute__((address_space(1))) u32x4_t* fm_gvec3_t;

template <int WM, int WN, int OUT_T, bool A_KMAJOR>
__device__ __forceinline__ void x3_rows_layer(const FusedNet& net, const FusedLayer& L, int M, int m0, float* act, int lda,
                                              int row_tile0, int col_tile0) {
  using namespace fm;
  const int lane = threadIdx.x & 63;
  const int l31 = lane & 31, lh = lane >> 5;
  const int K = L.K, N = L.N;
  const int NT32 = (N + 31) / 32;
  const int KS = (K + 15) / 16, KSq = KS & ~3;
  if (col_tile0 >= NT32) {
    if (OUT_T != 2) {
      __syncthreads();
      __syncthreads();
    }
    return;
  }
  unsigned b_voff[WN];
#pragma unroll
  for (int j = 0; j < WN; ++j)
    b_voff[j] = (unsigned)((col_tile0 + j < NT32 ? col_tile0 + j : NT32 - 1) * 3072 + lane * 16);
  fm_gbyte_t wb = (fm_gbyte_t)L.W;
  const float* a_rd = A_KMAJOR ? act + (lh * 8) * 64 + row_tile0 * 32 + l31 : act + (row_tile0 * 32 + l31) * lda + lh * 8;

  constexpr int R = WN >= 5 ? 3 : 4;                     // weight ring: R - 1 k-steps in flight + the one being multiplied
  f32x16 acc[WM][WN];
  f32x4 ra[2][WM][2];
  Pieces ap[2][WM];
  u32x4_t fb[R][WN][3];
  constexpr int NDS = A_KMAJOR ? 8 * WM : 2 * WM;        // LDS reads of a k-step
  auto aread1 = [&](int ks, f32x4 (&a)[WM][2], int r) {   // LDS read r of k-step ks
    if (A_KMAJOR) {
      const int i = r / 8, e = r % 8;
      a[i][e / 4][e % 4] = a_rd[(ks * 16 + e) * 64 + i * 32];
    } else {
      a[r / 2][r % 2] = *reinterpret_cast<const f32x4*>(a_rd + (r / 2) * 32 * lda + ks * 16 + (r % 2) * 4);
    }
  };
  auto aread = [&](int ks, f32x4 (&a)[WM][2]) {
    const int kc = ks < KS ? ks : KS - 1;
#pragma unroll
    for (int r = 0; r < NDS; ++r) aread1(kc, a, r);
  };
  auto bload = [&](u32x4_t (&b)[WN][3], int ks) {
    const int kc = ks < KS ? ks : KS - 1;
    fm_gbyte_t p = wb + (size_t)kc * NT32 * 3072;
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int q = 0; q < 3; ++q) b[j][q] = *(fm_gvec3_t)(p + b_voff[j] + q * 1024);
  };
  auto split = [&](const f32x4 (&a)[WM][2], Pieces (&q)[WM]) {
#pragma unroll
    for (int i = 0; i < WM; ++i) q[i] = split8(a[i][0][0], a[i][0][1], a[i][0][2], a[i][0][3], a[i][1][0], a[i][1][1], a[i][1][2], a[i][1][3]);
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
  auto step = [&](int s, const Pieces (&ap_cur)[WM], Pieces (&ap_nxt)[WM], f32x4 (&ra_nxt)[WM][2], f32x4 (&ra_free)[WM][2],
                  const u32x4_t (&b_cur)[WN][3], u32x4_t (&b_free)[WN][3]) {
    constexpr int NM = WM * WN * 6, NP = WM * 4, NMEM = 3 * WN + NDS, NCH = NP;
    const int kb = s + R - 1 < KS ? s + R - 1 : KS - 1, ka = s + 2 < KS ? s + 2 : KS - 1;
    fm_gbyte_t pb = wb + (size_t)kb * NT32 * 3072;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
#pragma unroll
      for (int o = c * NMEM / NCH; o < (c + 1) * NMEM / NCH; ++o) {
        if (o < 3 * WN) b_free[o / 3][o % 3] = *(fm_gvec3_t)(pb + b_voff[o / 3] + (o % 3) * 1024);
        else aread1(ka, ra_free, o - 3 * WN);
      }
      {
        const int i = c / 4, q = c % 4;
        const f32x4& v = ra_nxt[i][q / 2];
        unsigned h, m, l;
        split_pair(v[(q % 2) * 2], v[(q % 2) * 2 + 1], h, m, l);
        ap_nxt[i].p[0][q] = h; ap_nxt[i].p[1][q] = m; ap_nxt[i].p[2][q] = l;
      }
#pragma unroll
      for (int mm = c * NM / NCH; mm < (c + 1) * NM / NCH; ++mm) {
        const int t = mm / (WM * WN), i = (mm / WN) % WM, j = mm % WN;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ap_cur[i].p[X3_PA[t]]),
                                                            __builtin_bit_cast(bf16x8_t, b_cur[j][X3_PB[t]]), acc[i][j], 0, 0, 0);
      }
#pragma unroll
      for (int mm = 0; mm < (NM + NCH - 1) / NCH; ++mm) { FM_SGB(SG_MFMA, 1); FM_SGB(0x002, 2); if (mm < 3) FM_SGB(SG_VMEM_RD | SG_DS_RD, 1); }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // Pipelined part: whole periods of the (A parity x weight ring) rotation -- 4 steps with the four-slot ring, 6 with the
  // three-slot ring of the wide instantiation (five column tiles per wave: a fourth slot of 60 registers spilled).
  constexpr int P = R == 4 ? 4 : 6;
  const int KSp = KS / P * P;
  if (KSp) {
    bload(fb[0], 0);
    bload(fb[1], 1);
    if constexpr (R == 4) bload(fb[2], 2);
    aread(0, ra[0]);
    aread(1, ra[1]);
    split(ra[0], ap[0]);
    for (int g = 0; g < KSp; g += P) {
      if constexpr (R == 4) {
        step(g, ap[0], ap[1], ra[1], ra[0], fb[0], fb[3]);
        step(g + 1, ap[1], ap[0], ra[0], ra[1], fb[1], fb[0]);
        step(g + 2, ap[0], ap[1], ra[1], ra[0], fb[2], fb[1]);
        step(g + 3, ap[1], ap[0], ra[0], ra[1], fb[3], fb[2]);
      } else {
        step(g, ap[0], ap[1], ra[1], ra[0], fb[0], fb[2]);
        step(g + 1, ap[1], ap[0], ra[0], ra[1], fb[1], fb[0]);
        step(g + 2, ap[0], ap[1], ra[1], ra[0], fb[2], fb[1]);
        step(g + 3, ap[1], ap[0], ra[0], ra[1], fb[0], fb[2]);
        step(g + 4, ap[0], ap[1], ra[1], ra[0], fb[1], fb[0]);
        step(g + 5, ap[1], ap[0], ra[0], ra[1], fb[2], fb[1]);
      }
    }
  }
  // the steps behind the whole periods (K = 200: one of 13), one after the other: read, split, load, multiply
  for (int t = KSp; t < KS; ++t) {
    aread(t, ra[0]);
    bload(fb[0], t);
    split(ra[0], ap[0]);
    mma(ap[0], fb[0]);
  }

  if (OUT_T == 2) {
    float* dst = net.out + (size_t)(m0 >> 6) * net.ld_out * 64;
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int n = (col_tile0 + j) * 32 + l31;
      if (col_tile0 + j >= NT32 || n >= net.ld_out) continue;
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = (row_tile0 + i) * 32 + 8 * q + 4 * lh;
          *reinterpret_cast<f32x4*>(dst + (size_t)n * 64 + row) =
              f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
        }
    }
    return;
  }
  __syncthreads();   // every wave has read its last A values: the buffer may be overwritten
  static_assert(OUT_T == 1 || OUT_T == 2, "x3_rows_layer writes C^T to LDS or the tile-layout output");
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    if (col_tile0 + j >= NT32) continue;
    const int n = (col_tile0 + j) * 32 + l31;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (row_tile0 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        act[n * 64 + ((row + n) & 63)] = acc[i][j][r];
      }
  }
  __syncthreads();
}

__global__ __launch_bounds__(fm::NT) void mlp_fused_kernel(FusedMlpArgs args) {
  using namespace fm;
  extern __shared__ __attribute__((aligned(16))) float act[];
  const FusedNet& net = args.net[blockIdx.y];
  const int M = args.M, m0 = blockIdx.x * BM;
  const int tid = threadIdx.x;

  // ---- the network input: rows m0 .. m0+63 of x, columns [0, K0), zero up to the padded k-group boundary
  {
    const int K0 = net.layer[0].K;
    const int kpad = (((K0 + 7) / 8 + 3) & ~3) * 8;   // multiple of 32
    const int c4n = kpad / 4;                          // 16-byte pieces per row
    for (int i = tid; i < BM * c4n; i += NT) {
      const int r = i / c4n, c = (i % c4n) * 4;
      const int row = m0 + r < M ? m0 + r : M - 1;     // rows past the end repeat the last one (never stored)
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
    if (L.N <= 32) fused_layer_t<1, 1>(net, L, M, m0, act, LDA, wave & 1, wave >> 1, l, last);          // 32 x 32, two waves
    else if (L.N <= 128) fused_layer_t<1, 2>(net, L, M, m0, act, LDA, wave & 1, (wave >> 1) * 2, l, last);   // 32 x 64 per wave
    else fused_layer_t<2, 4>(net, L, M, m0, act, LDA, 0, wave * 4, l, last);                            // 64 x 128 per wave
  }
}

// A single linear layer with the same machinery, for contractions whose K fits the LDS with the row block
// (the SMPL blend-shape GEMMs, K = 200 / 320): the A block [BM][K] is staged once, the weights stream from L2 in
// fragment order, no barrier in the K loop.  Waves are arranged WROWS x WCOLS, each WM x WN tiles of 32 x 32.
template <int BM_, int WM, int WN, int WCOLS>
__global__ __launch_bounds__(fm::NT) void gemm_rows_kernel(FusedMlpArgs args) {
  extern __shared__ __attribute__((aligned(16))) float act[];
  const FusedNet& net = args.net[0];
  const FusedLayer& L = net.layer[0];
  const int M = args.M, m0 = blockIdx.x * BM_;
  const int tid = threadIdx.x;
  const int K0 = L.K;
  const RowsLds lay = rows_lds(K0, BM_, false);
  const int kpad = lay.kpad, lda = lay.lda;
  {
    const int c4n = kpad / 4;
    for (int i0 = tid; i0 < BM_ * c4n; i0 += 4 * fm::NT) {   // four 16-byte pieces per thread in flight
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * fm::NT;
        v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (i < BM_ * c4n) {
          const int r = i / c4n, c = (i % c4n) * 4;
          const int row = m0 + r < M ? m0 + r : M - 1;
          if (c < K0) v[u] = *reinterpret_cast<const f32x4*>(net.x + (size_t)row * net.ldx + c);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * fm::NT;
        if (i < BM_ * c4n) *reinterpret_cast<f32x4*>(act + (i / c4n) * lda + (i % c4n) * 4) = v[u];
      }
    }
  }
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  fused_layer_t<WM, WN>(net, L, M, m0, act, lda, (wave / WCOLS) * WM, (wave % WCOLS) * WN, 0, true);
}

template <int BM_, int WM, int WN, int WCOLS>
static hipError_t launch_gemm_rows_cfg(const FusedMlpArgs& args, hipStream_t stream) {
  const size_t lds = rows_lds(args.net[0].layer[0].K, BM_, false).bytes();
  if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(gemm_rows_kernel<BM_, WM, WN, WCOLS>), lds)) return e;
  hipLaunchKernelGGL((gemm_rows_kernel<BM_, WM, WN, WCOLS>), dim3((args.M + BM_ - 1) / BM_), dim3(fm::NT), lds, stream,
                     args);
  return hipGetLastError();
}

// C[M][N] = A[M][K] . Wp^T with Wp in fragment order. Returns false when the row-block kernel does not pay (the caller
// then uses the generic GEMM).  Measured at M = 32768: N = 320, K = 200 as 2 x 2 waves of 64 x 160 over 128 rows:
// 57 us against 73 us for the generic tile; N = 200, K = 320 as 2 x 2 waves of 32 x 128 over 64 rows: 65 us against 55,
// so only the first configuration is dispatched.
bool gemm_rows_applicable(int M, int N, int K) {
  if (K % 4 != 0) return false;
  return N <= 320 && N > 256 && K <= 288 && M >= 128 * 192;
}

hipError_t launch_gemm_rows(const float* A, int lda, const float* Wp, float* C, int ldc, int M, int N, int K,
                            hipStream_t stream) {
  FusedMlpArgs a;
  a.count = 1; a.M = M;
  FusedNet& fn = a.net[0];
  fn.x = A; fn.ldx = lda; fn.out = C; fn.ld_out = ldc; fn.n_layers = 1;
  FusedLayer& L = fn.layer[0];
  L.W = Wp; L.K = K; L.N = N; L.scale = nullptr; L.shift = nullptr; L.slope = 0.f; L.act = 0;
  if (N > 256) return launch_gemm_rows_cfg<128, 2, 5, 2>(a, stream);
  return launch_gemm_rows_cfg<64, 1, 4, 2>(a, stream);
}

// The row-block GEMM with tile-layout operands (kernels.h "tile layout"; the frame-per-lane SMPL kernel, smpl_tile.hip,
// reads and writes columns of 64 frames): 64 rows (= one tile) per workgroup, 2 x 2 waves of 32 x (32 WN).  A comes
// row-major or in tile layout (A_T: copied to LDS as it lies, K-major); C leaves straight from the accumulators as
// 16-byte pieces of the tile-layout columns.  No LDS beyond the A block: two workgroups per CU, so that one's staging
// and stores run beside the other's K loop (with C^T staged in LDS, 82 KB, it was one).
namespace rt {
// A_t[tile][k][64] -> act[k][64] as it lies (fused_layer_t's K-major operand): 16-byte pieces, no transposition; the
// columns [K0, kpad) are zero
__device__ __forceinline__ void stage_a_tile(const float* x_t, int ldx, int tile, int K0, int kpad, float* act) {
  const int tid = threadIdx.x;
  const f32x4* src = reinterpret_cast<const f32x4*>(x_t + (size_t)tile * ldx * 64);
  for (int i0 = tid; i0 < kpad * 16; i0 += 4 * fm::NT) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * fm::NT;
      v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (i < kpad * 16 && (i >> 4) < K0) v[u] = src[i];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * fm::NT;
      if (i < kpad * 16) reinterpret_cast<f32x4*>(act)[i] = v[u];
    }
  }
}
// row-major A[M][ldx] -> act[64][lda] (rows past the end repeat the last one; never stored)
__device__ __forceinline__ void stage_a_rows(const float* x, int ldx, int m0, int M, int K0, int kpad, int lda, float* act) {
  const int tid = threadIdx.x;
  const int c4n = kpad / 4;
  for (int i0 = tid; i0 < 64 * c4n; i0 += 4 * fm::NT) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * fm::NT;
      v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (i < 64 * c4n) {
        const int r = i / c4n, c = (i % c4n) * 4;
        const int row = m0 + r < M ? m0 + r : M - 1;
        if (c < K0) v[u] = *reinterpret_cast<const f32x4*>(x + (size_t)row * ldx + c);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * fm::NT;
      if (i < 64 * c4n) *reinterpret_cast<f32x4*>(act + (i / c4n) * lda + (i % c4n) * 4) = v[u];
    }
  }
}
// C^T in LDS ([column][64], rows rotated by the column; fused_layer_t<.., OUT_T>) out as 16-byte pieces of whole columns
__device__ __forceinline__ void store_ct(const float* act, int N, float* out_t, int ld_out, int tile) {
  const int NT32 = (N + 31) / 32;
  float* dst = out_t + (size_t)tile * ld_out * 64;
  for (int i = threadIdx.x; i < ld_out * 16; i += fm::NT) {
    const int n = i >> 4, f = (i & 15) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (n < NT32 * 32) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = act[n * 64 + ((f + e + n) & 63)];
    }
    *reinterpret_cast<f32x4*>(dst + (size_t)n * 64 + f) = v;
  }
}
}  // namespace rt

template <int WN, bool A_T>
__global__ __launch_bounds__(fm::NT) void gemm_rows_t_kernel(FusedMlpArgs args) {
  extern __shared__ __attribute__((aligned(16))) float act[];
  const FusedNet& net = args.net[0];
  const FusedLayer& L = net.layer[0];
  const int M = args.M, tile = blockIdx.x, m0 = tile * 64;
  const int K0 = L.K;
  const RowsLds lay = rows_lds(K0, 64, A_T);
  const int kpad = lay.kpad, lda = lay.lda;
  if (A_T) rt::stage_a_tile(net.x, net.ldx, tile, K0, kpad, act);
  else rt::stage_a_rows(net.x, net.ldx, m0, M, K0, kpad, lda, act);
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  fused_layer_t<1, WN, 2, A_T>(net, L, M, m0, act, lda, wave >> 1, (wave & 1) * WN, 0, true);
}

// The blend-shape GEMM of the frame-per-lane path with the iteration's pose / shape update, Rodrigues and the feature
// row as its PROLOGUE (feat_rows.h: the body of update_feat_kernel): the 64 x 200 feature block is built in LDS from
// 76 floats per frame and never exists in HBM.  out_t = feat . Wc2^T in tile layout.
constexpr int BLEND_FEAT_K = 200;   // feature columns (feat_rows.h); a tile of 64 frames touches at most 64 windows
__host__ __device__ constexpr RowsLds blend_feat_lds() { return rows_lds(BLEND_FEAT_K, 64, false, 0, 64 * 10); }
template <int WN, bool X3 = false>
__global__ __launch_bounds__(fm::NT) void blend_feat_gemm_kernel(FusedMlpArgs args, FeatArgs fa) {
  if (X3) X3_EXCLUSIVE_SIMD();
  extern __shared__ __attribute__((aligned(16))) float act[];
  const FusedNet& net = args.net[0];
  const FusedLayer& L = net.layer[0];
  const int M = args.M, tile = blockIdx.x, m0 = tile * 64;
  const int tid = threadIdx.x;
  constexpr int K0 = BLEND_FEAT_K;
  constexpr RowsLds lay = blend_feat_lds();
  constexpr int lda = lay.lda;
  for (int i = tid; i < 64 * (lda - K0); i += fm::NT) act[(i / (lda - K0)) * lda + K0 + i % (lda - K0)] = 0.f;
  const int fl = tid >> 5, slot = tid & 31;
  // window means of the shape update, once per window of the tile (32 lanes per window), through the pad columns'
  // neighbours in LDS: s_mean[window in tile][10] behind the A block
  float* s_mean = act + lay.extra_off;   // [<= 64 windows of the tile][10]
  const bool avg = fa.d_beta && fa.shape_avg;
  const int t_last = min(m0 + 63, M - 1);
  const int w_first = m0 / fa.F, n_win = avg ? t_last / fa.F - w_first + 1 : 0;   // <= 64
  for (int w = fl; w < n_win; w += 8) {
    const float d = shape_window_mean(fa, w_first + w, slot);
    if (slot >= NB) s_mean[w * 10 + slot - NB] = d;
  }
  if (avg) __syncthreads();
  // 32 lanes per frame, eight frames per pass: the loads of all eight passes first, then Rodrigues and the rows
  FeatLane v[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int t = m0 + p * 8 + fl;
    const float d = (avg && slot >= NB && t < M) ? s_mean[(t / fa.F - w_first) * 10 + slot - NB] : 0.f;
    v[p] = feat_lane_load(fa, t, slot, d);
  }
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int r = p * 8 + fl;
    if (m0 + r >= M) {
      for (int c = slot; c < K0; c += 32) act[r * lda + c] = 0.f;
    }
    feat_lane_finish<true>(fa, m0 + r, slot, v[p], act + r * lda, nullptr);
  }
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (X3) x3_rows_layer<1, WN, 2, false>(net, L, M, m0, act, lda, wave >> 1, (wave & 1) * WN);
  else fused_layer_t<1, WN, 2>(net, L, M, m0, act, lda, wave >> 1, (wave & 1) * WN, 0, true);
}

// The transposed blend-shape GEMM of the frame-per-lane path (d_feat = d_out . Wc2, both in tile layout) with the
// Rodrigues reverse as its EPILOGUE (feat_rows.h: the body of rodrigues_bwd_t_kernel): the feature cotangents stay in
// LDS, what leaves is g_theta / g_beta in the caller's rows.
constexpr int BLEND_T_N = 200;   // feature cotangent columns; rodrigues_bwd_tile's staging: TL_FR x 77 floats (feat_rows.h)
__host__ __device__ constexpr RowsLds blend_t_rod_lds(int K) {
  return rows_lds(K, 64, true, BLEND_T_N, TL_FR * ROD_BWD_LD, true);
}
template <int WN, bool X3 = false>
__global__ __launch_bounds__(fm::NT) void blend_t_gemm_rod_kernel(FusedMlpArgs args, RodBwdTArgs ra) {
  if (X3) X3_EXCLUSIVE_SIMD();
  extern __shared__ __attribute__((aligned(16))) float act[];
  const FusedNet& net = args.net[0];
  const FusedLayer& L = net.layer[0];
  const int M = args.M, tile = blockIdx.x, m0 = tile * 64;
  const int K0 = L.K;
  const RowsLds lay = blend_t_rod_lds(K0);
  rt::stage_a_tile(net.x, net.ldx, tile, K0, lay.kpad, act);
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (X3) x3_rows_layer<1, WN, 1, true>(net, L, M, m0, act, lay.lda, wave >> 1, (wave & 1) * WN);
  else fused_layer_t<1, WN, 1, true>(net, L, M, m0, act, lay.lda, wave >> 1, (wave & 1) * WN, 0, true);
  float* sg = act + lay.extra_off;   // behind C^T (the A block is dead by now)
  const int lane = threadIdx.x & 63;
  rodrigues_bwd_tile<true>(ra, tile, sg, [&](int col) { return act[col * 64 + ((lane + col) & 63)]; });
}

template <int WN, bool A_T>
static hipError_t launch_gemm_rows_t_cfg(const FusedMlpArgs& args, hipStream_t stream) {
  const size_t lds = rows_lds(args.net[0].layer[0].K, 64, A_T).bytes();
  if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(gemm_rows_t_kernel<WN, A_T>), lds)) return e;
  hipLaunchKernelGGL((gemm_rows_t_kernel<WN, A_T>), dim3((args.M + 63) / 64), dim3(fm::NT), lds, stream, args);
  return hipGetLastError();
}

hipError_t launch_gemm_rows_t(const float* A, int lda, bool a_tile, const float* Wp, float* C_t, int ldc_t, int M, int N,
                              int K, hipStream_t stream) {
  if (K % 4 != 0 || N > 320 || ldc_t != ((N + 31) / 32) * 32) return hipErrorInvalidValue;
  FusedMlpArgs a;
  a.count = 1; a.M = M;
  FusedNet& fn = a.net[0];
  fn.x = A; fn.ldx = lda; fn.out = C_t; fn.ld_out = ldc_t; fn.n_layers = 1;
  FusedLayer& L = fn.layer[0];
  L.W = Wp; L.K = K; L.N = N; L.scale = nullptr; L.shift = nullptr; L.slope = 0.f; L.act = 0;
  if (N > 256) return a_tile ? launch_gemm_rows_t_cfg<5, true>(a, stream) : launch_gemm_rows_t_cfg<5, false>(a, stream);
  return a_tile ? launch_gemm_rows_t_cfg<4, true>(a, stream) : launch_gemm_rows_t_cfg<4, false>(a, stream);
}

// The two initial-estimate heads (reference models.py:511-526: pose 66 + shape 10 columns on the LSTM output) as ONE
// product over the 76 stacked columns: the 64 x K block of y is staged once (the two-problem launch on the 256 x 128 tile
// read y twice and multiplied 256 padded columns), the result leaves through LDS to its two destinations -- the pose
// columns of the network input rows and the shape buffer -- with the bias added on the way.
struct HeadsArgs {
  const float* bias;        // [76]
  float* theta; int ld_theta;
  float* shape; int ld_shape;
  int n_pose, n_all;        // 66, 76
};
template <bool X3>
__global__ __launch_bounds__(fm::NT) void heads_rows_kernel(FusedMlpArgs args, HeadsArgs h) {
  if (X3) X3_EXCLUSIVE_SIMD();
  extern __shared__ __attribute__((aligned(16))) float act[];
  const FusedNet& net = args.net[0];
  const FusedLayer& L = net.layer[0];
  const int M = args.M, m0 = blockIdx.x * 64;
  const int K0 = L.K;
  const RowsLds lay = rows_lds(K0, 64, false, h.n_all);
  rt::stage_a_rows(net.x, net.ldx, m0, M, K0, lay.kpad, lay.lda, act);
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (X3) x3_rows_layer<1, 2, 1, false>(net, L, M, m0, act, lay.lda, wave & 1, (wave >> 1) * 2);
  else fused_layer_t<1, 2, 1>(net, L, M, m0, act, lay.lda, wave & 1, (wave >> 1) * 2, 0, true);
  // C^T in LDS ([column][64], rows rotated by the column)
  for (int i = threadIdx.x; i < 64 * h.n_all; i += fm::NT) {
    const int row = i / h.n_all, n = i - row * h.n_all;
    if (m0 + row >= M) continue;
    const float v = act[n * 64 + ((row + n) & 63)] + h.bias[n];
    if (n < h.n_pose) h.theta[(size_t)(m0 + row) * h.ld_theta + n] = v;
    else h.shape[(size_t)(m0 + row) * h.ld_shape + (n - h.n_pose)] = v;
  }
}

static FusedMlpArgs rows_t_args(const float* A, int lda, const float* Wp, float* C_t, int ldc_t, int M, int N, int K) {
  FusedMlpArgs a;
  a.count = 1; a.M = M;
  FusedNet& fn = a.net[0];
  fn.x = A; fn.ldx = lda; fn.out = C_t; fn.ld_out = ldc_t; fn.n_layers = 1;
  FusedLayer& L = fn.layer[0];
  L.W = Wp; L.K = K; L.N = N; L.scale = nullptr; L.shift = nullptr; L.slope = 0.f; L.act = 0;
  return a;
}

// Kernels that issue v_mfma_f32_32x32x16_bf16 run ONE wave per SIMD (scripts/dev/bf16_hazard_repro.md: with two, single
// accumulator elements come out wrong).  Their workgroups are four waves, so one workgroup per CU is the rule, enforced BY
// CONSTRUCTION here: the dynamic LDS request is raised above half a CU's LDS, so a second workgroup never fits -- whatever
// the kernel's own footprint (33 KB for the init heads of a 128-wide LSTM) and register count would have allowed (round 6;
// before, only the 512-wide layouts were safe, by their size).
static size_t x3_exclusive_lds(size_t lds) {
  const size_t half = LDS_BYTES_PER_CU / 2 + 1024;
  return lds > half ? lds : half;
}

template <class Kern, class Extra>
static hipError_t launch_rows_t_fused(Kern kern, size_t lds, const FusedMlpArgs& a, const Extra& x, hipStream_t stream) {
  if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return e;
  hipLaunchKernelGGL(kern, dim3((a.M + 63) / 64), dim3(fm::NT), lds, stream, a, x);
  return hipGetLastError();
}

// `x3`: Wp holds three bf16 pieces per weight (api.hip pack_fragments_x3_raw) and the product runs on the bf16 matrix path
hipError_t launch_blend_feat_gemm(const FeatArgs& fa, const float* Wp, float* C_t, int ldc_t, int N, bool x3,
                                  hipStream_t stream) {
  if (N > 320 || ldc_t != ((N + 31) / 32) * 32) return hipErrorInvalidValue;
  const FusedMlpArgs a = rows_t_args(nullptr, 0, Wp, C_t, ldc_t, fa.T, N, BLEND_FEAT_K);
  const size_t lds = blend_feat_lds().bytes();   // the A block + the tile's window means
  if (x3) {
    if (N > 256) return launch_rows_t_fused(blend_feat_gemm_kernel<5, true>, x3_exclusive_lds(lds), a, fa, stream);
    return launch_rows_t_fused(blend_feat_gemm_kernel<4, true>, x3_exclusive_lds(lds), a, fa, stream);
  }
  if (N > 256) return launch_rows_t_fused(blend_feat_gemm_kernel<5>, lds, a, fa, stream);
  return launch_rows_t_fused(blend_feat_gemm_kernel<4>, lds, a, fa, stream);
}

hipError_t launch_blend_t_gemm_rod(const float* A_t, int lda_t, const float* Wp, int K, const RodBwdTArgs& ra, bool x3,
                                   hipStream_t stream) {
  if (K % 4 != 0) return hipErrorInvalidValue;
  const FusedMlpArgs a = rows_t_args(A_t, lda_t, Wp, nullptr, 0, ra.T, BLEND_T_N, K);
  if (x3) return launch_rows_t_fused(blend_t_gemm_rod_kernel<4, true>, x3_exclusive_lds(blend_t_rod_lds(K).bytes()), a, ra, stream);
  return launch_rows_t_fused(blend_t_gemm_rod_kernel<4>, blend_t_rod_lds(K).bytes(), a, ra, stream);
}

bool heads_rows_applicable(int M, int K) { return M >= 4096 && K % 4 == 0 && K <= FUSED_MAX_WIDTH; }

hipError_t launch_heads_rows(const float* y, int ldy, const float* Wp, const float* bias, float* theta, int ld_theta,
                             float* shape, int ld_shape, int M, int K, int n_pose, int n_shape, bool x3, hipStream_t stream) {
  if (n_pose + n_shape > 128) return hipErrorInvalidValue;
  const FusedMlpArgs a = rows_t_args(y, ldy, Wp, nullptr, 0, M, n_pose + n_shape, K);
  HeadsArgs h{bias, theta, ld_theta, shape, ld_shape, n_pose, n_pose + n_shape};
  // the staged rows of y, later the transposed result (whole 32-column tiles): whichever is larger (a narrow LSTM's
  // row block is smaller than the 96 x 64 result) -- rows_lds, the layout the kernel indexes with
  const size_t lds = rows_lds(K, 64, false, n_pose + n_shape).bytes();
  if (x3) {
    const size_t lds1 = x3_exclusive_lds(lds);
    if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(heads_rows_kernel<true>), lds1)) return e;
    hipLaunchKernelGGL(heads_rows_kernel<true>, dim3((M + 63) / 64), dim3(fm::NT), lds1, stream, a, h);
    return hipGetLastError();
  }
  if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(heads_rows_kernel<false>), lds)) return e;
  hipLaunchKernelGGL(heads_rows_kernel<false>, dim3((M + 63) / 64), dim3(fm::NT), lds, stream, a, h);
  return hipGetLastError();
}

hipError_t launch_mlp_fused(const FusedMlpArgs& args, hipStream_t stream) {
  if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(mlp_fused_kernel), fm::LDS_BYTES)) return e;
  dim3 grid((args.M + fm::BM - 1) / fm::BM, args.count);
  hipLaunchKernelGGL(mlp_fused_kernel, grid, dim3(fm::NT), fm::LDS_BYTES, stream, args);
  return hipGetLastError();
}

}  // namespace empose
