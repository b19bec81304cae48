// A whole update MLP (reference nn/layers.py:46-77: Linear-BN-PReLU, hidden blocks, Linear; eval mode) in ONE launch.
//
// A workgroup owns 128 batch rows of one net and takes them through every layer: the four waves split the (<= 512)
// output columns of a layer into 128-column quarters, so a wave's tile is 128 x 128 = 4 x 4 MFMA tiles (256 accumulator
// registers, one wave per SIMD), exactly the wave tile of gemm_wide_f32_kernel.  Because the workgroup holds complete
// rows, layer l+1 only needs what this same workgroup wrote for layer l: the activations make a round trip through the
// L2 (own rows, ping-pong scratch) and a workgroup barrier -- no launch boundary, no grid-wide dependency, no HBM read
// of the activations, and the narrow first (K = 296) and last (N = 66 / 10) layers ride along instead of paying their
// own badly shaped launches.
//
// Per layer the K loop is software-pipelined by hand like the wide GEMM, with 16-wide K tiles so that the double-
// buffered LDS holds the 128 + 512 operand rows (2 x 51,200 B; rows padded to 20 floats: conflict-free ds_read_b128):
//   group A (k 0..7 : 64 MFMAs) | fragment reads of k 8..15 | LDS writes of K tile j+1 (fetched during tile j)
//   barrier
//   group B (k 8..15: 64 MFMAs) | fragment reads of k 0..7 of tile j+1 | global loads of K tile j+2
#include "gemm_epilogue.h"

#include <type_traits>

namespace empose {

namespace fm {
constexpr int BM = 128, BN = 512, BK = 16, LDT = BK + 4, NT = 256;
constexpr int STAGE = (BM + BN) * LDT;
constexpr size_t LDS_BYTES = 2 * (size_t)STAGE * sizeof(float);
constexpr int SG_MFMA = 0x008, SG_VMEM_RD = 0x020, SG_DS_RD = 0x100, SG_DS_WR = 0x200;
}  // namespace fm

#define FM_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)

#ifdef EMPOSE_FUSED_TRACE   // dev lab only: shader-clock stamps of block (0,0): per layer start, loop start, loop end, end
__device__ long long g_fused_trace[64];
__device__ int g_fused_layer;
#define FM_STAMP(i) if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) g_fused_trace[(i)] = clock64();
#else
#define FM_STAMP(i)
#endif

typedef const __attribute__((address_space(1))) f32x4* fm_gvec_t;
typedef const __attribute__((address_space(1))) char* fm_gbyte_t;

// One layer for the workgroup's 128 rows.  NARROW (N <= 128, the output layers): the four waves split the ROWS
// (32 each, all <= 128 columns) instead of the columns, so a 66- or 10-column layer does not cost a 512-column one.
template <bool NARROW>
__device__ __forceinline__ void fused_layer(const FusedNet& net, const FusedLayer& L, int M, int m0, float* lds,
                                            int layer_index) {
  FM_STAMP(4 * layer_index)
  using namespace fm;
  constexpr int WM = NARROW ? 1 : 4;          // 32-row tiles per wave
  constexpr int NW = NARROW ? 2 : 8;          // 64-row pieces of the weight tile this thread moves
  constexpr int NG = 2 + NW;
  constexpr int NMMA = WM * 4 * 4;            // MFMAs per k-group of 8
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int K = L.K, N = L.N;
  const float* A = L.in_buf < 0 ? net.x : net.buf[L.in_buf];
  const int lda = L.in_buf < 0 ? net.ldx : net.ld_buf;
  const float* W = L.W;

  // Global side: thread t moves 16 bytes of row (t / 4) + 64 i, columns 4 (t % 4) .. +3 of the K tile.
  const int lr = tid >> 2, lc = (tid & 3) * 4;
  unsigned a_off[2], w_off[NW];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = m0 + lr + 64 * i;
    a_off[i] = (unsigned)(((long)(r < M ? r : M - 1) * lda + lc) * 4);
  }
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    const int r = lr + 64 * i;
    w_off[i] = (unsigned)(((long)(r < N ? r : N - 1) * K + lc) * 4);
  }
  const int wofs = lr * LDT + lc;                                       // LDS write offset of piece 0; piece i: + 64 rows
  const int a_rd = ((NARROW ? wave * 32 : 0) + l31) * LDT + lh * 4;     // A fragments: row tile i adds 32 rows
  const int b_rd = (BM + (NARROW ? 0 : wave * 128) + l31) * LDT + lh * 4;

  f32x16 acc[WM][4];
  f32x4 g[NG];
  f32x4 fa[2][WM], fb[2][4];
  bool g_ok = true;
  const int nk = (K + BK - 1) / BK;
  const bool ragged_k = (K % BK) != 0;

  auto fread = [&](const float* st, int kk, f32x4 (&a)[WM], f32x4 (&b)[4]) {
#pragma unroll
    for (int i = 0; i < WM; ++i) a[i] = *reinterpret_cast<const f32x4*>(st + a_rd + i * 32 * LDT + kk * 8);
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const f32x4*>(st + b_rd + j * 32 * LDT + kk * 8);
  };
  auto lwrite = [&](float* st) {
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(st + wofs + i * 64 * LDT) = g[i];
#pragma unroll
    for (int i = 0; i < NW; ++i) *reinterpret_cast<f32x4*>(st + BM * LDT + wofs + i * 64 * LDT) = g[2 + i];
  };
  auto mma = [&](const f32x4 (&a)[WM], const f32x4 (&b)[4]) {   // consecutive MFMAs go to different accumulator tiles
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
  };
  auto gload_to = [&](f32x4 (&dst)[NG], int kt) -> bool {   // K tile kt (clamped past the end: fetched, never used)
    const int kc = kt < nk ? kt : nk - 1;
    const bool ok = kc * BK + lc < K;
    const unsigned back = ok ? 0u : (unsigned)(lc * 4);   // lanes past K re-read chunk 0 of the tile; zeroed later
    fm_gbyte_t pa = (fm_gbyte_t)(A + kc * BK);
    fm_gbyte_t pw = (fm_gbyte_t)(W + kc * BK);
#pragma unroll
    for (int i = 0; i < 2; ++i) dst[i] = *(fm_gvec_t)(pa + (a_off[i] - back));
#pragma unroll
    for (int i = 0; i < NW; ++i) dst[2 + i] = *(fm_gvec_t)(pw + (w_off[i] - back));
    return ok;
  };
  auto gload = [&](int kt) { g_ok = gload_to(g, kt); };
  auto gzero = [&]() {
#pragma unroll
    for (int i = 0; i < NG; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) g[i][e] = g_ok ? g[i][e] : 0.f;
  };
  // interleaving of one k-group: R fragment reads, then X other memory operations, spread over the group's MFMAs
  auto pattern = [&](auto other) {
    constexpr int other_mask = decltype(other)::value;
    constexpr int R = WM + 4;
    constexpr int per_r = NARROW ? 1 : 2, per_x = NARROW ? 2 : 4;
#pragma unroll
    for (int q = 0; q < R; ++q) { FM_SGB(SG_MFMA, per_r); FM_SGB(SG_DS_RD, 1); }
#pragma unroll
    for (int q = 0; q < NG; ++q) { FM_SGB(SG_MFMA, per_x); __builtin_amdgcn_sched_group_barrier(other_mask, 1, 0); }
    FM_SGB(SG_MFMA, NMMA - R * per_r - NG * per_x);
  };

#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- prologue: K tiles 0 and 1 are requested together (one round trip), tile 0 -> stage 0
  {
    f32x4 g1[NG];
    gload(0);
    const bool ok1 = gload_to(g1, 1);
    if (ragged_k && nk == 1) gzero();
    lwrite(lds);
#pragma unroll
    for (int i = 0; i < NG; ++i) g[i] = g1[i];
    g_ok = ok1;
  }
  __syncthreads();
  fread(lds, 0, fa[0], fb[0]);
  FM_STAMP(4 * layer_index + 1)

  for (int kt = 0; kt < nk; ++kt) {
    const float* cur = lds + (kt & 1) * STAGE;
    float* nxt = lds + ((kt + 1) & 1) * STAGE;
    if (ragged_k && kt + 2 == nk) gzero();   // uniform branch: the registers hold the (ragged) last K tile
    // ---- group A
    fread(cur, 1, fa[1], fb[1]);
    lwrite(nxt);
    mma(fa[0], fb[0]);
    pattern(std::integral_constant<int, SG_DS_WR>{});
    __syncthreads();
    // ---- group B
    fread(nxt, 0, fa[0], fb[0]);
    gload(kt + 2);
    mma(fa[1], fb[1]);
    pattern(std::integral_constant<int, SG_VMEM_RD>{});
  }

  FM_STAMP(4 * layer_index + 2)
  // ---- epilogue: scale/shift (bias, folded BatchNorm), PReLU, skip connection; to the scratch rows of this workgroup
  // or, for the last layer, to the net's output.  Waves whose tile lies past the layer's width have nothing to store.
  GemmProb p;
  p.C = L.out_buf < 0 ? net.out : net.buf[L.out_buf];
  p.ldc = L.out_buf < 0 ? net.ld_out : net.ld_buf;
  p.M = M; p.N = N; p.K = K;
  p.scale = L.scale; p.shift = L.shift;
  p.resid = L.resid_buf < 0 ? nullptr : net.buf[L.resid_buf];
  p.ldr = net.ld_buf;
  p.act = L.act; p.slope = L.slope;
  p.A = nullptr; p.W = nullptr; p.lda = 0; p.ldw = 0;
  // (Staging the tile through LDS for 16-byte stores was measured slower: with every CU in its epilogue at the same
  // time the limit is the ~7 TB/s the chip absorbs, not the store instruction count.)
  if (NARROW) epilogue<WM, 4>(p, acc, m0 + wave * 32, 0, l31, lh);
  else if (wave * 128 < N) epilogue<WM, 4>(p, acc, m0, wave * 128, l31, lh);
  // The next layer reads these rows back (this workgroup only): stores drained, then the barrier.  All waves share the
  // CU's vector L1, which the stores wrote through.
  __syncthreads();
  FM_STAMP(4 * layer_index + 3)
}

__global__ __launch_bounds__(fm::NT) void mlp_fused_kernel(FusedMlpArgs args) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const FusedNet& net = args.net[blockIdx.y];
  const int m0 = blockIdx.x * fm::BM;
  for (int l = 0; l < net.n_layers; ++l) {
    const FusedLayer& L = net.layer[l];
    if (L.N <= 128) fused_layer<true>(net, L, args.M, m0, lds, l);
    else fused_layer<false>(net, L, args.M, m0, lds, l);
  }
}

hipError_t launch_mlp_fused(const FusedMlpArgs& args, hipStream_t stream) {
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_fused_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)fm::LDS_BYTES);
    if (e != hipSuccess) return e;
    attr = true;
  }
  dim3 grid((args.M + fm::BM - 1) / fm::BM, args.count);
  hipLaunchKernelGGL(mlp_fused_kernel, grid, dim3(fm::NT), fm::LDS_BYTES, stream, args);
  return hipGetLastError();
}

}  // namespace empose
