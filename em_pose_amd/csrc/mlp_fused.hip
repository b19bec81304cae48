// A whole update MLP (reference nn/layers.py:46-77: Linear-BN-PReLU, hidden blocks, Linear; eval mode) in ONE launch.
//
// A workgroup owns 64 batch rows of one net and takes them through every layer.  Because it holds complete rows,
// layer l+1 only needs what this same workgroup wrote for layer l: the activations make a round trip through the L2
// (own rows, ping-pong scratch; 128 KB per workgroup and layer, so they stay L2-resident) and a workgroup barrier --
// no launch boundary, no grid-wide dependency, no HBM read of the activations; the narrow first (K = 296) and last
// (N = 66 / 10) layers ride along instead of paying their own badly shaped launches.
//
// Operands.  The four waves split a layer's (<= 512) output columns into 128-column quarters: a wave's tile is
// 64 x 128 = 2 x 4 MFMA tiles (128 accumulator registers), which leaves room for TWO workgroups per CU -- their
// layers drift apart, so one's epilogue (a burst of stores the memory system absorbs at only ~7 TB/s chip-wide) and
// prologue hide behind the other's matrix work.
//   * weights: never touch LDS.  They are packed once (api.hip pack_fragments) in MFMA fragment order, so a wave's B
//     fragment of a k-group is one coalesced 1 KB global load straight into registers, prefetched three k-groups ahead
//     in a four-slot register ring (the weights of both nets are L2-resident).
//   * activations: the 64 x 32 K tile goes global -> registers -> LDS (double-buffered, rows padded to 36 floats:
//     conflict-free ds_read_b128), shared by the four waves.
// The K loop is software-pipelined by hand (K tile = four k-groups of 8; `sched_group_barrier` pins the interleaving):
//   group 0 : 32 MFMAs | A fragments of group 1 | B ring slot 3 <- (this tile, group 3)
//   group 1 : 32 MFMAs | A fragments of group 2 | B ring slot 0 <- (next tile, group 0)
//   group 2 : 32 MFMAs | A fragments of group 3 | B ring slot 1 <- (next tile, group 1) | LDS writes of the next A tile
//   barrier
//   group 3 : 32 MFMAs | A fragments of group 0 of the next tile | B ring slot 2 | global loads of the A tile after next
// Narrow layers (N <= 128, the output layers) split rows AND columns over the waves (32 x 64 each) so that a 66- or
// 10-column layer does not cost a 512-column one.
#include "gemm_epilogue.h"

#include <type_traits>

namespace empose {

namespace fm {
constexpr int BM = 64, BK = 32, LDT = BK + 4, NT = 256;
constexpr int STAGE = BM * LDT;
constexpr int SG_MFMA = 0x008, SG_VMEM_RD = 0x020, SG_DS_RD = 0x100, SG_DS_WR = 0x200;
}  // namespace fm

#define FM_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)

#ifdef EMPOSE_FUSED_TRACE   // dev lab only: shader-clock stamps of block (0,0): per layer start, loop start, loop end, end
__device__ long long g_fused_trace[64];
#define FM_STAMP(i) if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) g_fused_trace[(i)] = clock64();
#else
#define FM_STAMP(i)
#endif

typedef const __attribute__((address_space(1))) f32x4* fm_gvec_t;
typedef const __attribute__((address_space(1))) char* fm_gbyte_t;

// One layer for the workgroup's 64 rows.
template <bool NARROW>
__device__ __forceinline__ void fused_layer(const FusedNet& net, const FusedLayer& L, int M, int m0, float* lds,
                                            int layer_index) {
  using namespace fm;
  FM_STAMP(4 * layer_index)
  constexpr int WM = NARROW ? 1 : 2;          // 32-row tiles per wave
  constexpr int WN = NARROW ? 2 : 4;          // 32-column tiles per wave
  constexpr int NMMA = WM * WN * 4;           // MFMAs per k-group of 8
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int K = L.K, N = L.N;
  const float* A = L.in_buf < 0 ? net.x : net.buf[L.in_buf];
  const int lda = L.in_buf < 0 ? net.ldx : net.ld_buf;
  const int NT32 = (N + 31) / 32;             // column tiles of the packed weights
  const int KG = (K + 7) / 8;                 // k-groups of the packed weights
  const int row_tile0 = NARROW ? (wave & 1) : 0;
  const int col_tile0 = NARROW ? (wave >> 1) * 2 : wave * 4;
  // Column tiles past the layer's width are clamped to the last one: fetched and multiplied like the others (no
  // branch in the pipelined loop), never stored.
  unsigned b_tile[WN];
#pragma unroll
  for (int j = 0; j < WN; ++j) b_tile[j] = (unsigned)((col_tile0 + j < NT32 ? col_tile0 + j : NT32 - 1) * 1024);

  // ---- A side: thread t moves 16 bytes of row (t / 8) + 32 i (i = 0, 1), columns 4 (t % 8) .. +3 of the K tile
  const int lr = tid >> 3, lc = (tid & 7) * 4;
  unsigned a_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = m0 + lr + 32 * i;
    a_off[i] = (unsigned)(((long)(r < M ? r : M - 1) * lda + lc) * 4);
  }
  const int wofs = lr * LDT + lc;
  const int a_rd = (row_tile0 * 32 + l31) * LDT + lh * 4;
  const int nk = (K + BK - 1) / BK;
  const bool ragged_k = (K % BK) != 0;
  // ---- B side: byte offset of this lane inside a fragment; fragment (kg, nt) starts at ((kg * NT32 + nt) * 64) * 16
  fm_gbyte_t wb = (fm_gbyte_t)L.W + (size_t)lane * 16;

  f32x16 acc[WM][WN];
  f32x4 ga[2];                  // A staging (one K tile)
  bool ga_ok = true;
  f32x4 fa[2][WM];              // A fragments, double-buffered over the k-groups
  f32x4 fb[4][WN];              // B fragment ring: slot s holds k-group (4 t + s) of some K tile t

  auto fread = [&](const float* st, int kk, f32x4 (&a)[WM]) {
#pragma unroll
    for (int i = 0; i < WM; ++i) a[i] = *reinterpret_cast<const f32x4*>(st + a_rd + i * 32 * LDT + kk * 8);
  };
  auto bload = [&](f32x4 (&b)[WN], int kg) {   // k-group kg of the packed weights (clamped: fetched, never used)
    const int kc = kg < KG ? kg : KG - 1;
    fm_gbyte_t p = wb + (size_t)kc * NT32 * 1024;
#pragma unroll
    for (int j = 0; j < WN; ++j) b[j] = *(fm_gvec_t)(p + b_tile[j]);
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
  auto aload_to = [&](f32x4 (&dst)[2], int kt) -> bool {   // K tile kt of the activations (clamped past the end)
    const int kc = kt < nk ? kt : nk - 1;
    const bool ok = kc * BK + lc < K;
    const unsigned back = ok ? 0u : (unsigned)(lc * 4);   // lanes past K re-read chunk 0 of the tile; zeroed later
    fm_gbyte_t pa = (fm_gbyte_t)(A + kc * BK);
#pragma unroll
    for (int i = 0; i < 2; ++i) dst[i] = *(fm_gvec_t)(pa + (a_off[i] - back));
    return ok;
  };
  auto aload = [&](int kt) { ga_ok = aload_to(ga, kt); };
  auto azero = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) ga[i][e] = ga_ok ? ga[i][e] : 0.f;
  };
  auto awrite = [&](float* st) {
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(st + wofs + i * 32 * LDT) = ga[i];
  };
  // interleaving of one k-group: WM fragment reads, WN weight loads, X other memory operations among NMMA MFMAs
  auto pattern = [&](auto other, auto n_other) {
    constexpr int mask = decltype(other)::value, X = decltype(n_other)::value;
    constexpr int step = NARROW ? 1 : 2;
#pragma unroll
    for (int q = 0; q < WM; ++q) { FM_SGB(SG_MFMA, step); FM_SGB(SG_DS_RD, 1); }
#pragma unroll
    for (int q = 0; q < WN; ++q) { FM_SGB(SG_MFMA, step); FM_SGB(SG_VMEM_RD, 1); }
#pragma unroll
    for (int q = 0; q < X; ++q) { FM_SGB(SG_MFMA, step); __builtin_amdgcn_sched_group_barrier(mask, 1, 0); }
    FM_SGB(SG_MFMA, NMMA - step * (WM + WN + X));
  };
  using C0 = std::integral_constant<int, 0>;
  using C2 = std::integral_constant<int, 2>;
  using MDW = std::integral_constant<int, SG_DS_WR>;
  using MVR = std::integral_constant<int, SG_VMEM_RD>;

#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- prologue: weight ring slots 0..2, A tiles 0 (-> stage 0) and 1 (in flight)
  bload(fb[0], 0);
  bload(fb[1], 1);
  bload(fb[2], 2);
  {
    f32x4 g1[2];
    aload(0);
    const bool ok1 = aload_to(g1, 1);
    if (ragged_k && nk == 1) azero();
    awrite(lds);
    ga[0] = g1[0]; ga[1] = g1[1]; ga_ok = ok1;
  }
  __syncthreads();
  fread(lds, 0, fa[0]);
  FM_STAMP(4 * layer_index + 1)

  for (int kt = 0; kt < nk; ++kt) {
    const float* cur = lds + (kt & 1) * STAGE;
    float* nxt = lds + ((kt + 1) & 1) * STAGE;
    if (ragged_k && kt + 2 == nk) azero();   // uniform branch: the registers hold the (ragged) last A tile
    // ---- group 0
    fread(cur, 1, fa[1]);
    bload(fb[3], kt * 4 + 3);
    mma(fa[0], fb[0]);
    pattern(MDW{}, C0{});
    // ---- group 1
    fread(cur, 2, fa[0]);
    bload(fb[0], kt * 4 + 4);
    mma(fa[1], fb[1]);
    pattern(MDW{}, C0{});
    // ---- group 2
    fread(cur, 3, fa[1]);
    bload(fb[1], kt * 4 + 5);
    awrite(nxt);
    mma(fa[0], fb[2]);
    pattern(MDW{}, C2{});
    __syncthreads();
    // ---- group 3
    fread(nxt, 0, fa[0]);
    bload(fb[2], kt * 4 + 6);
    aload(kt + 2);
    mma(fa[1], fb[3]);
    pattern(MVR{}, C2{});
  }

  FM_STAMP(4 * layer_index + 2)
  // ---- epilogue: scale/shift (bias, folded BatchNorm), PReLU, skip connection; to the scratch rows of this workgroup
  // or, for the last layer, to the net's output.  Tiles past the layer's width are skipped by the column guard.
  GemmProb p;
  p.C = L.out_buf < 0 ? net.out : net.buf[L.out_buf];
  p.ldc = L.out_buf < 0 ? net.ld_out : net.ld_buf;
  p.M = M; p.N = N; p.K = K;
  p.scale = L.scale; p.shift = L.shift;
  p.resid = L.resid_buf < 0 ? nullptr : net.buf[L.resid_buf];
  p.ldr = net.ld_buf;
  p.act = L.act; p.slope = L.slope;
  p.A = nullptr; p.W = nullptr; p.lda = 0; p.ldw = 0;
  if (col_tile0 < NT32) epilogue<WM, WN>(p, acc, m0 + row_tile0 * 32, col_tile0 * 32, l31, lh);
  // The next layer reads these rows back (this workgroup only): stores drained, then the barrier.  All waves share the
  // CU's vector L1, which the stores wrote through.
  __syncthreads();
  FM_STAMP(4 * layer_index + 3)
}

__global__ __launch_bounds__(fm::NT, 2) void mlp_fused_kernel(FusedMlpArgs args) {
  __shared__ __attribute__((aligned(16))) float lds[2 * fm::STAGE];
  const FusedNet& net = args.net[blockIdx.y];
  const int m0 = blockIdx.x * fm::BM;
  for (int l = 0; l < net.n_layers; ++l) {
    const FusedLayer& L = net.layer[l];
    if (L.N <= 128) fused_layer<true>(net, L, args.M, m0, lds, l);
    else fused_layer<false>(net, L, args.M, m0, lds, l);
  }
}

hipError_t launch_mlp_fused(const FusedMlpArgs& args, hipStream_t stream) {
  dim3 grid((args.M + fm::BM - 1) / fm::BM, args.count);
  hipLaunchKernelGGL(mlp_fused_kernel, grid, dim3(fm::NT), 0, stream, args);
  return hipGetLastError();
}

}  // namespace empose
