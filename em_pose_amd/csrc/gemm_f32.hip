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

namespace empose {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
constexpr int LDT = BK + 4;  // padded LDS row (floats)

template <int ROWS>
__device__ __forceinline__ void load_tile(const float* __restrict__ base, int ld, int row0, int nrows, int k0, int K,
                                          int tid, float4 (&regs)[ROWS / 32]) {
#pragma unroll
  for (int i = 0; i < ROWS / 32; ++i) {
    const int slot = tid + i * 256;
    const int r = slot >> 3, c4 = (slot & 7) * 4;
    const int gr = row0 + r, gk = k0 + c4;
    if (gr < nrows && gk < K) {
      regs[i] = *reinterpret_cast<const float4*>(base + (size_t)gr * ld + gk);
    } else {
      regs[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

template <int ROWS>
__device__ __forceinline__ void store_tile(float* lds, int tid, const float4 (&regs)[ROWS / 32]) {
#pragma unroll
  for (int i = 0; i < ROWS / 32; ++i) {
    const int slot = tid + i * 256;
    const int r = slot >> 3, c4 = (slot & 7) * 4;
    *reinterpret_cast<float4*>(lds + r * LDT + c4) = regs[i];
  }
}

// ROLE only separates instantiations by name so that profilers report the update-net hidden layers (ROLE 1) apart
// from the other users of the same tile configuration.
template <int WM, int WN, int ROLE>
__global__ __launch_bounds__(256) void gemm_tn_f32_kernel(GemmBatch batch) {
  constexpr int BM = 64 * WM, BN = 64 * WN;
  __shared__ __attribute__((aligned(16))) float lds[(BM + BN) * LDT];
  float* As = lds;
  float* Bs = lds + BM * LDT;

  const GemmProb& p = batch.p[blockIdx.y];
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  if ((int)blockIdx.x >= tiles_n * tiles_m) return;
  const int m0 = ((int)blockIdx.x / tiles_n) * BM;
  const int n0 = ((int)blockIdx.x % tiles_n) * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wrow = wave >> 1, wcol = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[BM / 32], rb[BN / 32];
  const int nk = (p.K + BK - 1) / BK;
  load_tile<BM>(p.A, p.lda, m0, p.M, 0, p.K, tid, ra);
  load_tile<BN>(p.W, p.ldw, n0, p.N, 0, p.K, tid, rb);
  store_tile<BM>(As, tid, ra);
  store_tile<BN>(Bs, tid, rb);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) {
      load_tile<BM>(p.A, p.lda, m0, p.M, (kt + 1) * BK, p.K, tid, ra);
      load_tile<BN>(p.W, p.ldw, n0, p.N, (kt + 1) * BK, p.K, tid, rb);
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
    __syncthreads();
    if (kt + 1 < nk) {
      store_tile<BM>(As, tid, ra);
      store_tile<BN>(Bs, tid, rb);
      __syncthreads();
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
        if (p.act == 1) y = y >= 0.f ? y : p.slope * y;
        if (p.resid) y += p.resid[(size_t)m * p.ldr + n];
        p.C[(size_t)m * p.ldc + n] = y;
      }
    }
  }
}

template <int WM, int WN, int ROLE = 0>
static hipError_t launch_cfg(const GemmBatch& batch, hipStream_t stream) {
  constexpr int BM = 64 * WM, BN = 64 * WN;
  int blocks = 0;
  for (int i = 0; i < batch.count; ++i) {
    const GemmProb& p = batch.p[i];
    const int t = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    blocks = t > blocks ? t : blocks;
  }
  if (blocks == 0) return hipSuccess;
  hipLaunchKernelGGL((gemm_tn_f32_kernel<WM, WN, ROLE>), dim3(blocks, batch.count), dim3(256), 0, stream, batch);
  return hipGetLastError();
}

hipError_t launch_gemm(const GemmBatch& batch, hipStream_t stream) {
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
  if (shortm && narrow) return launch_cfg<1, 1>(batch, stream);
  if (shortm) return launch_cfg<1, 2>(batch, stream);
  if (narrow) return launch_cfg<2, 1>(batch, stream);
  if (nblocks(128, 128) >= 512) return batch.role == 1 ? launch_cfg<2, 2, 1>(batch, stream) : launch_cfg<2, 2>(batch, stream);
  if (nblocks(64, 128) >= 256) return launch_cfg<1, 2>(batch, stream);
  return launch_cfg<1, 1>(batch, stream);
}

}  // namespace empose
