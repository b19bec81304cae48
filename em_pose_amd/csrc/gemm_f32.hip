// fp32 linear layers on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, 64 FLOP/clk/SIMD).
//
// Every dense contraction of the LGD path goes through this kernel: the LSTM input projections, the init heads, the
// two 6-layer update MLPs (reference nn/layers.py:46-77, eval-mode BatchNorm folded into a per-column scale/shift and
// PReLU in the epilogue), and the blend-shape / joint-regression matrix of the SMPL sub-mesh (and its transpose for
// the reverse pass).  Operands are "A[M][K] row-major" x "W[N][K] row-major" (the nn.Linear weight layout), i.e.
// both K-contiguous, so both tiles are staged through LDS with the same 16-byte loads and read back as
// ds_read_b128; since a dot product does not care about the order of k, each half-wave takes 4 consecutive k of an
// 8-wide group, which maps the b128 read straight onto four 32x32x2 MFMAs.
//
// Tile: 256 threads = 2x2 waves, each wave WM x WN tiles of 32x32 -> block tile (64*WM) x (64*WN), BK = 32.
// LDS rows are padded to 36 floats: 16 lanes of a ds_read_b128 group then hit 16 distinct 16-byte slots.
#include "kernels.h"

#include <cstdlib>

namespace empose {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Tile configuration: WR x WC waves per block, each wave WM x WN tiles of 32x32, K tile BK, DB = LDS double buffering
// (one barrier per K tile instead of two).
template <int WR_, int WC_, int WM_, int WN_, int BK_, bool DB_>
struct Cfg {
  static constexpr int WR = WR_, WC = WC_, WM = WM_, WN = WN_, BK = BK_;
  static constexpr bool DB = DB_;
  static constexpr int NT = 64 * WR * WC;
  static constexpr int BM = 32 * WM * WR, BN = 32 * WN * WC;
  static constexpr int LDT = BK + 4;  // padded LDS row (floats): ds_read_b128 lane groups hit 16 distinct 16-byte slots
  static constexpr int C4 = BK / 4;   // float4 per tile row
  static constexpr int NA = BM * C4 / NT, NB_ = BN * C4 / NT;
  static constexpr int STAGE = (BM + BN) * LDT;
  static constexpr int LDS_FLOATS = STAGE * (DB ? 2 : 1);
  static_assert(BM * C4 % NT == 0 && BN * C4 % NT == 0, "tile rows must divide evenly over the threads");
};

template <typename C, int N>
__device__ __forceinline__ void load_tile(const float* __restrict__ base, int ld, int row0, int nrows, int k0, int K,
                                          int tid, float4 (&regs)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int slot = tid + i * C::NT;
    const int r = slot / C::C4, c4 = (slot % C::C4) * 4;
    const int gr = row0 + r, gk = k0 + c4;
    if (gr < nrows && gk < K) {
      regs[i] = *reinterpret_cast<const float4*>(base + (size_t)gr * ld + gk);
    } else {
      regs[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

template <typename C, int N>
__device__ __forceinline__ void store_tile(float* lds, int tid, const float4 (&regs)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int slot = tid + i * C::NT;
    const int r = slot / C::C4, c4 = (slot % C::C4) * 4;
    *reinterpret_cast<float4*>(lds + r * C::LDT + c4) = regs[i];
  }
}

// ROLE only separates instantiations by name so that profilers report the update-net hidden layers (ROLE 1) apart
// from the other users of the same tile configuration.
template <typename C, int ROLE>
__global__ __launch_bounds__(C::NT) void gemm_tn_f32_kernel(GemmBatch batch) {
  constexpr int BM = C::BM, BN = C::BN, BK = C::BK, LDT = C::LDT, WM = C::WM, WN = C::WN;
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const GemmProb& p = batch.p[blockIdx.y];
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int nt = tiles_n * tiles_m;
  if ((int)blockIdx.x >= nt) return;
  // XCD-aware tile order: workgroup b runs on XCD b % 8 (each XCD has its own L2), so hand every XCD a contiguous
  // run of tiles; the tiles_n column tiles that share one A row panel then hit the same L2.  Bijective for any nt.
  int tile = blockIdx.x;
  if (batch.xcd_swizzle) {
    const int q = nt / 8, r = nt % 8, xcd = tile % 8, k = tile / 8;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int m0 = (tile / tiles_n) * BM;
  const int n0 = (tile % tiles_n) * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wrow = wave / C::WC, wcol = wave % C::WC;
  const int l31 = lane & 31, lh = lane >> 5;

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[C::NA], rb[C::NB_];
  const int nk = (p.K + BK - 1) / BK;
  load_tile<C>(p.A, p.lda, m0, p.M, 0, p.K, tid, ra);
  load_tile<C>(p.W, p.ldw, n0, p.N, 0, p.K, tid, rb);
  store_tile<C>(lds, tid, ra);
  store_tile<C>(lds + BM * LDT, tid, rb);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const float* As = lds + (C::DB ? (kt & 1) * C::STAGE : 0);
    const float* Bs = As + BM * LDT;
    if (kt + 1 < nk) {
      load_tile<C>(p.A, p.lda, m0, p.M, (kt + 1) * BK, p.K, tid, ra);
      load_tile<C>(p.W, p.ldw, n0, p.N, (kt + 1) * BK, p.K, tid, rb);
    }
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      float4 a[WM], b[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i)
        a[i] = *reinterpret_cast<const float4*>(As + (wrow * 32 * WM + i * 32 + l31) * LDT + kk * 8 + lh * 4);
#pragma unroll
      for (int j = 0; j < WN; ++j)
        b[j] = *reinterpret_cast<const float4*>(Bs + (wcol * 32 * WN + j * 32 + l31) * LDT + kk * 8 + lh * 4);
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
        }
    }
    if (C::DB) {
      // The other stage was last read in iteration kt-1, and every wave has passed that iteration's barrier.
      if (kt + 1 < nk) {
        float* nx = lds + ((kt + 1) & 1) * C::STAGE;
        store_tile<C>(nx, tid, ra);
        store_tile<C>(nx + BM * LDT, tid, rb);
      }
      __syncthreads();
    } else {
      __syncthreads();
      if (kt + 1 < nk) {
        store_tile<C>(lds, tid, ra);
        store_tile<C>(lds + BM * LDT, tid, rb);
        __syncthreads();
      }
    }
  }

  // Epilogue. C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const int n = n0 + wcol * 32 * WN + j * 32 + l31;
    if (n >= p.N) continue;
    const float sc = p.scale ? p.scale[n] : 1.f;
    const float sh = p.shift ? p.shift[n] : 0.f;
#pragma unroll
    for (int i = 0; i < WM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wrow * 32 * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m >= p.M) continue;
        float y = acc[i][j][r] * sc + sh;
        if (p.act == 2) {  // residual block of the ResNet baseline: relu(W x + b + x), reference layers.py:170-182
          if (p.resid) y += p.resid[(size_t)m * p.ldr + n];
          y = y > 0.f ? y : 0.f;
        } else {
          if (p.act == 1) y = y >= 0.f ? y : p.slope * y;
          if (p.resid) y += p.resid[(size_t)m * p.ldr + n];  // skip connection around a block (after the activation)
        }
        p.C[(size_t)m * p.ldc + n] = y;
      }
    }
  }
}

template <typename C, int ROLE = 0>
static hipError_t launch_cfg(const GemmBatch& batch, hipStream_t stream) {
  int blocks = 0;
  for (int i = 0; i < batch.count; ++i) {
    const GemmProb& p = batch.p[i];
    const int t = ((p.M + C::BM - 1) / C::BM) * ((p.N + C::BN - 1) / C::BN);
    blocks = t > blocks ? t : blocks;
  }
  if (blocks == 0) return hipSuccess;
  constexpr size_t lds = (size_t)C::LDS_FLOATS * sizeof(float);
  static bool attr = false;
  if (!attr && lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_f32_kernel<C, ROLE>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr = true;
  }
  hipLaunchKernelGGL((gemm_tn_f32_kernel<C, ROLE>), dim3(blocks, batch.count), dim3(C::NT), lds, stream, batch);
  return hipGetLastError();
}

using CfgS11 = Cfg<2, 2, 1, 1, 32, false>;   //  64 x  64
using CfgS12 = Cfg<2, 2, 1, 2, 32, false>;   //  64 x 128
using CfgS21 = Cfg<2, 2, 2, 1, 32, false>;   // 128 x  64
using CfgL = Cfg<2, 2, 2, 2, 32, false>;     // 128 x 128, 2 barriers per K tile
using CfgLdb = Cfg<2, 2, 2, 2, 32, true>;    // 128 x 128, double-buffered LDS
using CfgX = Cfg<4, 2, 2, 2, 32, false>;     // 256 x 128, 8 waves
using CfgXdb = Cfg<4, 2, 2, 2, 32, true>;    // 256 x 128, 8 waves, double-buffered
using CfgL64 = Cfg<2, 2, 2, 2, 64, false>;   // 128 x 128, BK = 64
using CfgY = Cfg<2, 4, 2, 2, 32, false>;     // 128 x 256, 8 waves

template <int ROLE>
static hipError_t launch_large(const GemmBatch& batch, hipStream_t stream) {
  static const int variant = getenv("EMPOSE_GEMM_VARIANT") ? atoi(getenv("EMPOSE_GEMM_VARIANT")) : 0;  // dev A/B only
  // Measured on MI355X (M=65536, N=K=512): 128x128/4 waves 94 TF, +LDS double buffer 96, BK=64 95,
  // 256x128/8 waves 105, 128x256/8 waves 106 TFLOP/s -> the 8-wave 256x128 tile is the default.
  switch (variant) {
    case 1: return launch_cfg<CfgLdb, ROLE>(batch, stream);
    case 2: return launch_cfg<CfgL, ROLE>(batch, stream);
    case 3: return launch_cfg<CfgXdb, ROLE>(batch, stream);
    case 4: return launch_cfg<CfgL64, ROLE>(batch, stream);
    case 5: return launch_cfg<CfgY, ROLE>(batch, stream);
    default: return launch_cfg<CfgX, ROLE>(batch, stream);
  }
}

hipError_t launch_gemm(const GemmBatch& batch_in, hipStream_t stream) {
  GemmBatch batch = batch_in;
  static const int swz = getenv("EMPOSE_GEMM_SWIZZLE") ? atoi(getenv("EMPOSE_GEMM_SWIZZLE")) : 1;  // dev A/B only
  batch.xcd_swizzle = swz;
  int maxM = 0, maxN = 0;
  for (int i = 0; i < batch.count; ++i) {
    maxM = batch.p[i].M > maxM ? batch.p[i].M : maxM;
    maxN = batch.p[i].N > maxN ? batch.p[i].N : maxN;
  }
  // Narrow outputs (heads: 66 / 10 columns) and short batches take the smaller tiles; everything else 128x128,
  // unless that would leave most of the 256 CUs without a block.
  const bool narrow = maxN <= 64;
  const bool shortm = maxM <= 64;
  auto nblocks = [&](int bm, int bn) {
    long t = 0;
    for (int i = 0; i < batch.count; ++i)
      t += (long)((batch.p[i].M + bm - 1) / bm) * ((batch.p[i].N + bn - 1) / bn);
    return t;
  };
  if (shortm && narrow) return launch_cfg<CfgS11>(batch, stream);
  if (shortm) return launch_cfg<CfgS12>(batch, stream);
  if (narrow) return launch_cfg<CfgS21>(batch, stream);
  if (nblocks(128, 128) >= 512) return batch.role == 1 ? launch_large<1>(batch, stream) : launch_large<0>(batch, stream);
  if (nblocks(64, 128) >= 256) return launch_cfg<CfgS12>(batch, stream);
  return launch_cfg<CfgS11>(batch, stream);
}

}  // namespace empose
