// Full-mesh SMPL-H evaluation (reference smpl_torch_batch.py / bodymodels' SMPLLayer forward: v_posed = v_template +
// shapedirs . beta + posedirs . (R - I), then linear blend skinning of all V vertices), gfx950 only.
//
// One workgroup owns 64 frames for the whole launch: their 200 pose/shape features and their 22 relative bone
// transforms stay in LDS (120 KB), and the workgroup's eight waves walk the 32-vertex tiles of the mesh.  For a tile a
// wave computes v_posed of 32 vertices x 64 frames on the fp32 matrix cores -- six 32x32 accumulators: two frame
// tiles x the three coordinate planes, so a lane ends up with x, y and z of the same (frame, vertex) pairs -- and skins
// them in registers; the (T x 20670) v_posed matrix never exists in memory and the vertices are written once.
// The coefficient matrix is consumed from L2 in matrix-core fragment order (one coalesced 1 KB read per k-group and
// coordinate plane, packed by the host when the mesh handle is created) through a five-slot register ring that runs
// four k-groups ahead and straight across tile boundaries; there is no barrier after the staging.
// Measured (scripts/dev/mesh_lab.hip, T = 16384): a tile costs a wave 41.1k cycles of K loop (600 MFMAs, 38.4k ideal)
// plus 11.1k cycles of skinning when it has its SIMD to itself.  The second wave per SIMD hides load and LDS latency
// but does not overlap the two phases: fp32 vector FMAs and fp32 MFMAs share the SIMD's FMA lanes (the chip's vector
// and matrix fp32 peaks are the same number), so the cost is the sum of both, and the kernel runs at the power limit
// (~2.1 GHz).  The skinning is therefore written for the fewest vector instructions: packed FMAs on the halves of
// each 16-byte transform row.
#include <hip/hip_runtime.h>
#include <type_traits>

#include "bf16x3.h"
#include "gemm_epilogue.h"
#include "kernels.h"

namespace empose {

typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace mr {
constexpr int BM = 64;              // frames per workgroup
constexpr int NW = 8;               // waves per workgroup
constexpr int K = 200, KG = K / 8;  // 25 k-groups of 8
constexpr int LDA = K + 4;
constexpr int RING = 5;             // KG % RING == 0: a k-group's ring slot does not depend on the tile
constexpr int A_FLOATS = BM * LDA;
constexpr int XF_FLOATS = BM * NB * 12;
constexpr int TR_FLOATS = BM * 4;
constexpr size_t LDS_BYTES = (size_t)(A_FLOATS + XF_FLOATS + TR_FLOATS) * sizeof(float) + 64;
constexpr int TILE_FLOATS = KG * 3 * 256;   // one 32-vertex tile of the packed coefficients
static_assert(KG % RING == 0, "ring slots must line up across tiles");
}  // namespace mr

#ifdef EMPOSE_MESH_TRACE   // dev lab only (scripts/dev/mesh_lab.hip): shader-clock stamps of the waves of block (0,0)
__device__ long long g_mesh_trace[8 * 32 * 4];
#define MR_STAMP(tile, i) \
  if (lane == 0 && blockIdx.x == 0 && blockIdx.y == 0) g_mesh_trace[(wave * 32 + (tile)) * 4 + (i)] = clock64();
#else
#define MR_STAMP(tile, i)
#endif

#define MR_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0);
#define MR_MFMA 0x008
#define MR_VMEM_RD 0x020
#define MR_DS_RD 0x100

template <bool EXTRA>   // EXTRA: body models with more than four bones per vertex (not SMPL-H)
__global__ __launch_bounds__(mr::NW * 64) void mesh_rows_kernel(MeshSkinArgs a) {
  using namespace mr;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* As = lds;
  float* XFs = lds + A_FLOATS;
  float* TRs = XFs + XF_FLOATS;
  const int T = a.T, V = a.V;
  const int f0 = blockIdx.x * BM;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- staging: features, relative transforms and translations of the block's frames (rows past T repeat row T-1;
  // their results are never stored)
  {
    const float* __restrict__ feat = a.feat;
    for (int i = tid; i < BM * (K / 4); i += NW * 64) {
      const int r = i / (K / 4), c = (i % (K / 4)) * 4;
      const int row = f0 + r < T ? f0 + r : T - 1;
      *reinterpret_cast<f32x4*>(As + r * LDA + c) = *reinterpret_cast<const f32x4*>(feat + (size_t)row * K + c);
    }
    if (tid < BM) *reinterpret_cast<f32x4*>(As + tid * LDA + K) = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* __restrict__ xf = a.xf;
    for (int i = tid; i < BM * NB * 3; i += NW * 64) {
      const int r = i / (NB * 3), c = i % (NB * 3);
      const int row = f0 + r < T ? f0 + r : T - 1;
      *reinterpret_cast<f32x4*>(XFs + i * 4) = *reinterpret_cast<const f32x4*>(xf + ((size_t)row * NB * 3 + c) * 4);
    }
    if (tid < BM) {
      const int row = f0 + tid < T ? f0 + tid : T - 1;
      f32x4 t{0.f, 0.f, 0.f, 0.f};
      if (a.trans) { t[0] = a.trans[(size_t)row * 3]; t[1] = a.trans[(size_t)row * 3 + 1]; t[2] = a.trans[(size_t)row * 3 + 2]; }
      *reinterpret_cast<f32x4*>(TRs + tid * 4) = t;
    }
  }
  __syncthreads();

  // ---- this wave's tiles: vt = first + wave, + NW, ... below `end`
  const int n_tiles = (V + 31) / 32;
  const int per_block = (n_tiles + gridDim.y - 1) / gridDim.y;
  const int first = blockIdx.y * per_block;
  const int end = min(first + per_block, n_tiles);
  int vt = first + wave;
  if (vt >= end) return;

  epi_cgbyte_t wbase = (epi_cgbyte_t)a.wc_frag;
  const unsigned lane16 = (unsigned)lane * 16u;
  auto tile_base = [&](int t) { return wbase + (size_t)t * (TILE_FLOATS * 4); };
  auto bload = [&](f32x4 (&dst)[3], epi_cgbyte_t base, int g) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
      dst[c] = *(const __attribute__((address_space(1))) f32x4*)(base + ((unsigned)((g * 3 + c) * 1024) + lane16));
  };

  f32x4 ring[RING][3];
  {
    epi_cgbyte_t b0 = tile_base(vt);
#pragma unroll
    for (int g = 0; g < RING - 1; ++g) bload(ring[g], b0, g);
  }

  const float* a_lane = As + l31 * LDA + lh * 4;
  // A fragments: read one k-group ahead into three slots (k-groups u = 0..4 of a ring pass use slots 0,1,0,1,2, so the
  // slot being filled is never the one being consumed although a pass has an odd number of k-groups)
#define FA_SLOT(u) ((u) == RING - 1 ? 2 : ((u) & 1))
  f32x4 fa[3][2];
  fa[0][0] = *reinterpret_cast<const f32x4*>(a_lane);
  fa[0][1] = *reinterpret_cast<const f32x4*>(a_lane + 32 * LDA);
  epi_gbyte_t vbase = (epi_gbyte_t)a.vertices;
  const int kb = a.kb;
  const size_t vrow_bytes = (size_t)V * 12;

  for (int seq = 0; vt < end; vt += NW, ++seq) {
    MR_STAMP(seq, 0)
    epi_cgbyte_t bcur = tile_base(vt);
    // the prefetch runs into the wave's next tile; past the last one it re-reads this tile (never consumed)
    epi_cgbyte_t bnext = vt + NW < end ? tile_base(vt + NW) : bcur;
    const int s = vt * 32 + l31;            // this lane's vertex (the packed tables are padded to whole tiles)
    const int4 bone4 = *reinterpret_cast<const int4*>(a.skin_idx4 + (size_t)s * 4);
    const f32x4 w4 = *reinterpret_cast<const f32x4*>(a.skin_w4 + (size_t)s * 4);

    f32x16 acc[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][c][r] = 0.f;

#pragma unroll 1
    for (int g0 = 0; g0 < KG; g0 += RING) {
#pragma unroll
      for (int u = 0; u < RING; ++u) {
        const int g = g0 + u;
        {
          // A fragments of the next k-group (k-group 0 again after the last one: the next tile starts there)
          const int gn = g + 1 < KG ? g + 1 : 0;
          f32x4 (&fn)[2] = fa[u + 1 == RING ? 0 : FA_SLOT(u + 1)];
          fn[0] = *reinterpret_cast<const f32x4*>(a_lane + gn * 8);
          fn[1] = *reinterpret_cast<const f32x4*>(a_lane + 32 * LDA + gn * 8);
          const int gp = g + RING - 1;   // k-group to prefetch: this tile's, or the first ones of the next tile
          epi_cgbyte_t src = gp < KG ? bcur : bnext - (size_t)KG * 3 * 1024;
          bload(ring[(u + RING - 1) % RING], src, gp);
        }
        const f32x4 (&fc)[2] = fa[FA_SLOT(u)];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int c = 0; c < 3; ++c)
              acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(fc[i][e], ring[u][c][e], acc[i][c], 0, 0, 0);
        // 24 MFMAs with the two fragment reads and the three weight loads spread between them
        MR_SGB(MR_MFMA, 2) MR_SGB(MR_DS_RD, 1) MR_SGB(MR_MFMA, 2) MR_SGB(MR_DS_RD, 1)
        MR_SGB(MR_MFMA, 2) MR_SGB(MR_VMEM_RD, 1) MR_SGB(MR_MFMA, 2) MR_SGB(MR_VMEM_RD, 1)
        MR_SGB(MR_MFMA, 2) MR_SGB(MR_VMEM_RD, 1) MR_SGB(MR_MFMA, 14)
      }
    }

    MR_STAMP(seq, 1)
    // ---- skinning of the lane's 32 (frame, vertex) pairs: blended 3x4 transform, then one mat-vec
    // (reference order: T = sum_k w_k G_k, v = T . [v_posed, 1] + trans).  Addresses are a per-lane part that only
    // depends on the tile (bones, vertex, lane half) plus a compile-time / wave-uniform part per accumulator element.
    if (s < V) {
      const char* xfl = reinterpret_cast<const char*>(XFs) + lh * (4 * NB * 48);
      const char* xk[4] = {xfl + bone4.x * 48, xfl + bone4.y * 48, xfl + bone4.z * 48, xfl + bone4.w * 48};
      const f32x2 wp[4] = {{w4[0], w4[0]}, {w4[1], w4[1]}, {w4[2], w4[2]}, {w4[3], w4[3]}};
      const char* trl = reinterpret_cast<const char*>(TRs) + lh * 64;
      const unsigned lane_off = ((unsigned)(f0 + 4 * lh) * (unsigned)V + (unsigned)s) * 12u;
      auto skin = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int dm = i * 32 + (r & 3) + 8 * (r >> 2);   // frame within the block, less 4 * lh
            const float vx = acc[i][0][r], vy = acc[i][1][r], vz = acc[i][2][r];
            const f32x4 tr = *reinterpret_cast<const f32x4*>(trl + dm * 16);
            float out[3];
#pragma unroll
            for (int row = 0; row < 3; ++row) {
              // (T0, T1) and (T2, T3) as register pairs: two packed FMAs per bone on the halves of the 16-byte read
              f32x4 gk = *reinterpret_cast<const f32x4*>(xk[0] + dm * (NB * 48) + row * 16);
              f32x2 Ta = wp[0] * f32x2{gk[0], gk[1]}, Tb = wp[0] * f32x2{gk[2], gk[3]};
#pragma unroll
              for (int k = 1; k < 4; ++k) {
                gk = *reinterpret_cast<const f32x4*>(xk[k] + dm * (NB * 48) + row * 16);
                Ta = __builtin_elementwise_fma(wp[k], f32x2{gk[0], gk[1]}, Ta);
                Tb = __builtin_elementwise_fma(wp[k], f32x2{gk[2], gk[3]}, Tb);
              }
              if (EXTRA)
                for (int k = 4; k < kb; ++k) {
                  const int b = a.skin_idx[(size_t)s * kb + k];
                  const float wk = a.skin_w[(size_t)s * kb + k];
                  gk = *reinterpret_cast<const f32x4*>(xfl + b * 48 + dm * (NB * 48) + row * 16);
                  Ta = __builtin_elementwise_fma(f32x2{wk, wk}, f32x2{gk[0], gk[1]}, Ta);
                  Tb = __builtin_elementwise_fma(f32x2{wk, wk}, f32x2{gk[2], gk[3]}, Tb);
                }
              out[row] = __builtin_fmaf(Ta[0], vx, __builtin_fmaf(Ta[1], vy, __builtin_fmaf(Tb[0], vz, Tb[1]))) + tr[row];
            }
            if (FULL || f0 + 4 * lh + dm < T) {
              epi_gfloat_t o = (epi_gfloat_t)(vbase + (size_t)dm * vrow_bytes + lane_off);
              o[0] = out[0]; o[1] = out[1]; o[2] = out[2];
            }
          }
      };
      if (f0 + BM <= T) skin(std::true_type{}); else skin(std::false_type{});
    }
    MR_STAMP(seq, 2)
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The same kernel with the blend-shape contraction on the bf16 matrix cores ("split bf16", fp32 accumulate): an fp32
// operand x is the exact sum of bf16 pieces x = x0 + x1 + x2, |x1| <= 2^-8 |x|, |x2| <= 2^-16 |x|.  NOT the arithmetic
// of the headline: selected explicitly (empose_mesh_vertices_fwd_bf16x3, `bench.py --workload vertices --arith bf16x3`).
// Every column of the contraction carries a (hi, lo) pair on both operands and contributes three products
// hi.hi + hi.lo + lo.hi; which pieces sit in the pair depends on the column:
//   * pose blend-shapes (189 features, centimetre-scale contributions): (x0, x1) . (w0, w1) -- products down to 2^-16
//     relative, error < 1e-6 m;
//   * template + shape blend-shapes (10 betas and the constant 1, metre-scale) appear twice: once as (a0, a2) . (b0, b2)
//     and once as (a1, a0) . (b1, b0), which together are the six products a0b0 + a0b1 + a1b0 + a0b2 + a2b0 + a1b1 of a
//     three-piece split -- fp32-like.
// That is 211 columns = 14 k-steps of 16 (v_mfma_f32_32x32x16_bf16, 8 passes for 16 k where the fp32 instruction needs
// 16 passes for 2): the K loop of a tile shrinks from 600 fp32 MFMAs (38.4 k cycles) to 252 bf16 MFMAs (8.1 k).
// Measured (scripts/dev/mesh_bf16_lab.hip, T = 16384, one wave per SIMD): 13-14 k cycles of K loop (the coefficient
// stream, 84 KB per tile and wave from L2, three k-steps ahead in a register ring) and 12-13 k of skinning per tile,
// at 1.8-1.9 GHz under the power limit.
// ONE WAVE PER SIMD (four waves per workgroup), deliberately: with eight waves per workgroup (two per SIMD, as the fp32
// kernel runs) this kernel returned sporadically wrong x coordinates -- always the second accumulator row of the upper
// lane half, lanes 48..63, in roughly one tile-wave in a thousand, different ones on every launch; independent of the
// register count (216..254), of the prefetch, of fences / nops around the stores and after the K loop.  With one wave
// per SIMD ten repetitions of 16384 frames are bit-identical and within 5e-7 of the fp32 kernel.  Narrowed down with
// diagnostic builds: eight waves per workgroup of which four idle after the staging -> clean (it is two ACTIVE waves per
// SIMD, not the workgroup size); scalar instead of packed skinning arithmetic -> still wrong; the fp32 MFMA instruction
// in place of v_mfma_f32_32x32x16_bf16 (same kernel otherwise, two waves per SIMD) -> clean.  The throughput is
// the same either way: the two waves of a SIMD ran their K loops and their skinning in lockstep and gained nothing
// from each other.
// Operand order inside a k-step only has to agree between the two operands (a dot product is order-free): a lane's
// eight values are k = 16 * step + 8 * (lane >> 5) + 0..7 on both sides.
// ---------------------------------------------------------------------------------------------------------------
namespace mb {
#ifndef MB_NW
#define MB_NW 4
#endif
constexpr int BM = 64, NW = MB_NW;
constexpr int KS = 14;                    // k-steps: 189 pose columns + 2 x 11 shape columns = 211 <= 224
constexpr int M1 = 189, M2 = 200, KCOLS = 211;
constexpr int LDA = 232;                  // bf16 per row of a piece (464 B = 4 x 29 dwords: conflict-free 16-byte reads)
constexpr int A_PIECE_BYTES = BM * LDA * 2;
constexpr int A_BYTES = 2 * A_PIECE_BYTES;
constexpr int XF_FLOATS = BM * NB * 12;
constexpr int TR_FLOATS = BM * 4;
constexpr size_t LDS_BYTES = (size_t)A_BYTES + (size_t)(XF_FLOATS + TR_FLOATS) * sizeof(float) + 64;
constexpr int TILE_BYTES = KS * 3 * 2 * 1024;   // packed coefficients of one 32-vertex tile
#ifndef MB_RING
#define MB_RING 4
#endif
constexpr int RING = MB_RING;             // B fragments: steps s+1 .. s+RING-1 in flight while step s is consumed
static_assert(TILE_BYTES == MESH_BF16_TILE_BYTES, "api.hip packs what this kernel reads");
}  // namespace mb

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short bf16_rne(float x) {   // round to nearest even (finite inputs)
  const unsigned u = __float_as_uint(x);
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float bf16_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

template <bool EXTRA>
__global__ __launch_bounds__(mb::NW * 64) void mesh_rows_bf16_kernel(MeshSkinArgs a) {
  X3_EXCLUSIVE_SIMD();
  using namespace mb;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  unsigned short* Ab = reinterpret_cast<unsigned short*>(lds);
  float* XFs = reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + A_BYTES);
  float* TRs = XFs + XF_FLOATS;
  const int T = a.T, V = a.V;
  const int f0 = blockIdx.x * BM;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- staging: the features split into bf16 pieces (column plan above), relative transforms, translations
  {
    const float* __restrict__ feat = a.feat;
    for (int i = tid; i < BM * LDA; i += NW * 64) {
      const int r = i / LDA, c = i - r * LDA;
      const int row = f0 + r < T ? f0 + r : T - 1;
      const int src = c < M1 ? c : (c < M2 ? c : c - (M2 - M1));   // both shape groups read columns 189..199
      const float x = c < KCOLS ? feat[(size_t)row * 200 + src] : 0.f;
      const unsigned short p0 = bf16_rne(x);
      const float r1 = x - bf16_f32(p0);
      const unsigned short p1 = bf16_rne(r1);
      const unsigned short p2 = bf16_rne(r1 - bf16_f32(p1));
      Ab[i] = c < M2 ? p0 : p1;                                   // hi: x0 | a0 | a1
      Ab[A_PIECE_BYTES / 2 + i] = c < M1 ? p1 : (c < M2 ? p2 : p0);   // lo: x1 | a2 | a0
    }
    const float* __restrict__ xf = a.xf;
    for (int i = tid; i < BM * NB * 3; i += NW * 64) {
      const int r = i / (NB * 3), c = i % (NB * 3);
      const int row = f0 + r < T ? f0 + r : T - 1;
      *reinterpret_cast<f32x4*>(XFs + i * 4) = *reinterpret_cast<const f32x4*>(xf + ((size_t)row * NB * 3 + c) * 4);
    }
    if (tid < BM) {
      const int row = f0 + tid < T ? f0 + tid : T - 1;
      f32x4 t{0.f, 0.f, 0.f, 0.f};
      if (a.trans) { t[0] = a.trans[(size_t)row * 3]; t[1] = a.trans[(size_t)row * 3 + 1]; t[2] = a.trans[(size_t)row * 3 + 2]; }
      *reinterpret_cast<f32x4*>(TRs + tid * 4) = t;
    }
  }
  __syncthreads();

  const int n_tiles = (V + 31) / 32;
  const int per_block = (n_tiles + gridDim.y - 1) / gridDim.y;
  const int first = blockIdx.y * per_block;
  const int end = min(first + per_block, n_tiles);
  int vt = first + wave;
  if (vt >= end) return;

  // The coefficient table as a raw buffer: a fragment load is one buffer_load_dwordx4 with the lane offset in a VGPR, the
  // tile / k-step offset in an SGPR and the (plane, piece) offset in the instruction -- no address arithmetic in VGPRs
  // (flat global loads made the compiler keep a 64-bit address pair per 4 KB of the 84 KB tile).
  const __amdgpu_buffer_rsrc_t wtab = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(a.wc_bf16), 0, (int)((size_t)((V + 31) / 32) * TILE_BYTES), 0x00020000);
  const int lane16 = lane * 16;
  auto wload = [&](int tile_off, int ks, int c, int p) {
    const int f = c * 2 + p;   // fragment of the k-step: the part below 4 KB rides in the instruction's offset field
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wtab, lane16 + (f & 3) * 1024,
                                                                           tile_off + ks * 6144 + (f >> 2) * 4096, 0));
  };
  // B fragments of one k-step: [plane][piece], 1 KB each
  auto bload = [&](f32x4 (&dst)[3][2], int tile_off, int ks) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int p = 0; p < 2; ++p) dst[c][p] = wload(tile_off, ks, c, p);
  };
  auto tile_off_of = [&](int t) { return __builtin_amdgcn_readfirstlane(t) * TILE_BYTES; };   // wave-uniform
  // A fragments of this lane: row l31 (+ 32), 16 bytes at k-step * 32 + lh * 16; [frame tile][piece]
  const char* a_lane = reinterpret_cast<const char*>(Ab) + l31 * (LDA * 2) + lh * 16;
  auto aload = [&](f32x4 (&dst)[2][2], int ks) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int p = 0; p < 2; ++p)
        dst[i][p] = *reinterpret_cast<const f32x4*>(a_lane + p * A_PIECE_BYTES + i * 32 * (LDA * 2) + ks * 32);
  };
  epi_gbyte_t vbase = (epi_gbyte_t)a.vertices;
  const int kb = a.kb;
  const size_t vrow_bytes = (size_t)V * 12;

  f32x4 ring[RING][3][2];
  f32x4 fa[2][2][2];
  {
    const int b0 = tile_off_of(vt);
#pragma unroll
    for (int ks = 0; ks < RING - 1; ++ks) bload(ring[ks], b0, ks);
  }
  aload(fa[0], 0);

#pragma unroll 1
  for (int seq = 0; vt < end; vt += NW, ++seq) {
    MR_STAMP(seq, 0)
    const int bt = tile_off_of(vt);
    // past the wave's last tile the prefetch re-reads this one (never consumed)
    const int bnext = vt + NW < end ? tile_off_of(vt + NW) : bt;
    const int s = vt * 32 + l31;
    const int4 bone4 = *reinterpret_cast<const int4*>(a.skin_idx4 + (size_t)s * 4);
    const f32x4 w4 = *reinterpret_cast<const f32x4*>(a.skin_w4 + (size_t)s * 4);
    f32x16 acc[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][c][r] = 0.f;

#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      // Coefficients RING-1 k-steps ahead (every tile starts in slot 0: the first steps of the next tile are requested
      // after the loop and land during the skinning), A fragments one k-step ahead.  A k-step is three fenced groups of
      // six MFMAs, one per product, each on the six different accumulators: the scheduler must never put two MFMAs on
      // the same accumulator back to back (measured: it did, eight in a row, and besides being slow that gave
      // sporadically wrong rows of the accumulator on gfx950), so nothing but scalar / vector ALU work may cross a fence.
      const int kn = ks + RING - 1;
      f32x4 (&bn)[3][2] = ring[kn % RING];
      f32x4 (&fn)[2][2] = fa[(ks + 1) & 1];
      const int an = ks + 1 < KS ? ks + 1 : 0;
      const f32x4 (&fc)[2][2] = fa[ks & 1];
      const f32x4 (&bc)[3][2] = ring[ks % RING];
#pragma unroll
      for (int prod = 0; prod < 3; ++prod) {   // lo.hi, hi.lo, hi.hi: the small terms first
        // this group's share of the prefetch: one coordinate plane of coefficients, one frame tile of A fragments
        if (kn < KS) {
#pragma unroll
          for (int p = 0; p < 2; ++p)
            bn[prod][p] = wload(bt, kn, prod, p);
        }
        if (prod < 2) {
#pragma unroll
          for (int p = 0; p < 2; ++p)
            fn[prod][p] = *reinterpret_cast<const f32x4*>(a_lane + p * A_PIECE_BYTES + prod * 32 * (LDA * 2) + an * 32);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const bf16x8 av = __builtin_bit_cast(bf16x8, fc[i][prod == 0 ? 1 : 0]);
            const bf16x8 bv = __builtin_bit_cast(bf16x8, bc[c][prod == 1 ? 1 : 0]);
            acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[i][c], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0x6);   // only VALU / SALU may move across
      }
    }
#pragma unroll
    for (int ks = 0; ks < RING - 1; ++ks) bload(ring[ks], bnext, ks);

    MR_STAMP(seq, 1)
    // ---- skinning (as mesh_rows_kernel)
    if (s < V) {
      const char* xfl = reinterpret_cast<const char*>(XFs) + lh * (4 * NB * 48);
      const char* xk[4] = {xfl + bone4.x * 48, xfl + bone4.y * 48, xfl + bone4.z * 48, xfl + bone4.w * 48};
      const f32x2 wp[4] = {{w4[0], w4[0]}, {w4[1], w4[1]}, {w4[2], w4[2]}, {w4[3], w4[3]}};
      const char* trl = reinterpret_cast<const char*>(TRs) + lh * 64;
      const unsigned lane_off = ((unsigned)(f0 + 4 * lh) * (unsigned)V + (unsigned)s) * 12u;
      auto skin = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int dm = i * 32 + (r & 3) + 8 * (r >> 2);
            const float vx = acc[i][0][r], vy = acc[i][1][r], vz = acc[i][2][r];
            const f32x4 tr = *reinterpret_cast<const f32x4*>(trl + dm * 16);
            float out[3];
#pragma unroll
            for (int row = 0; row < 3; ++row) {
              f32x4 gk = *reinterpret_cast<const f32x4*>(xk[0] + dm * (NB * 48) + row * 16);
              f32x2 Ta = wp[0] * f32x2{gk[0], gk[1]}, Tb = wp[0] * f32x2{gk[2], gk[3]};
#pragma unroll
              for (int k = 1; k < 4; ++k) {
                gk = *reinterpret_cast<const f32x4*>(xk[k] + dm * (NB * 48) + row * 16);
                Ta = __builtin_elementwise_fma(wp[k], f32x2{gk[0], gk[1]}, Ta);
                Tb = __builtin_elementwise_fma(wp[k], f32x2{gk[2], gk[3]}, Tb);
              }
              if (EXTRA)
                for (int k = 4; k < kb; ++k) {
                  const int b = a.skin_idx[(size_t)s * kb + k];
                  const float wk = a.skin_w[(size_t)s * kb + k];
                  gk = *reinterpret_cast<const f32x4*>(xfl + b * 48 + dm * (NB * 48) + row * 16);
                  Ta = __builtin_elementwise_fma(f32x2{wk, wk}, f32x2{gk[0], gk[1]}, Ta);
                  Tb = __builtin_elementwise_fma(f32x2{wk, wk}, f32x2{gk[2], gk[3]}, Tb);
                }
              out[row] = __builtin_fmaf(Ta[0], vx, __builtin_fmaf(Ta[1], vy, __builtin_fmaf(Tb[0], vz, Tb[1]))) + tr[row];
            }
            if (FULL || f0 + 4 * lh + dm < T) {
              epi_gfloat_t o = (epi_gfloat_t)(vbase + (size_t)dm * vrow_bytes + lane_off);
              o[0] = out[0]; o[1] = out[1]; o[2] = out[2];
            }
          }
      };
      if (f0 + BM <= T) skin(std::true_type{}); else skin(std::false_type{});
    }
    MR_STAMP(seq, 2)
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Round 4: the same evaluation with the BONE BLEND as a second contraction on the matrix cores.
//
// In the kernel above the skinning costs more than the blend-shape contraction (12-13 k against 13-14 k cycles per tile):
// per (vertex, frame) it gathers four 3 x 4 bone transforms from LDS -- 12 x 16 bytes per lane and frame, 393 KB per
// tile and wave, i.e. it is bound by LDS bandwidth, not arithmetic.  But the blended transform is itself a product,
//   T[f][v][c] = sum_b W[v][b] G[f][b][c]      (c = 12 entries of the 3 x 4 transform, b = 22 bones, W the skin weights),
// with the same shape as the contraction before it: rows = frames, columns = the tile's 32 vertices, K = bones.  Laid out
// so, its C/D fragments line up with the blend-shape accumulators (lane = vertex, register = frame), and what is left for
// the vector unit is  out = T^R v + T^t  on registers: 9 FMAs per vertex and frame, no LDS access.
//   * W as a dense 32 (bones, 22 used) x 32 (vertices) block per tile, split hi + lo in bf16, in B-fragment order
//     (api.hip pack_mesh_skin_bf16): 4 KB per tile, prefetched with the coefficient ring;
//   * G of the workgroup's 64 frames staged ONCE as bf16 hi + lo pieces [piece][entry c][frame][24 bones] (48-byte rows:
//     conflict-free 16-byte fragment reads; the k-slots 24..31 of a fragment read into the next row -- finite values --
//     and meet zero weights), 72 KB next to the 58 KB feature block;
//   * three products per k-step (lo.hi, hi.lo, hi.hi) as for the blend shapes; what is dropped is 2^-18 relative per factor.
// 144 MFMAs + 32 x 12 FMAs replace 32 x (12 LDS reads + 57 packed FMAs) per tile and lane.
// MEASURED (T = 16384, V = 6890): correct (6e-6 of the fp32 kernel; the dropped lo.lo terms), and SLOWER than the vector
// skinning it replaces -- 14.7 M frames/s with the MFMAs and the apply one after the other, 15.1 M software-pipelined,
// 15.6 M with the translation folded into the accumulator, against 17.1 M for mesh_rows_bf16_kernel.  The tile then
// carries 396 instead of 252 v_mfma_f32_32x32x16_bf16; at the ~54-64 cycles each of them takes here (13-14 k cycles for
// the 252 of the K loop) the 144 added ones cost what the LDS-bound gather cost: padding 22 bones to a K of 32 and three
// products per k-step make this contraction too expensive for its 0.7 MFLOP of useful work.  Opt-in ("mesh_skin_mfma").
// One wave per SIMD, MFMAs in fenced groups on distinct accumulators, as above (the two-waves-per-SIMD corruption of the
// bf16 MFMA is still unexplained: scripts/dev/bf16_hazard_repro.md).
// ---------------------------------------------------------------------------------------------------------------
namespace ms {
using namespace mb;
constexpr int G_ROW_BYTES = 48;                                   // 24 bones x bf16
constexpr int G_PIECE_BYTES = 12 * BM * G_ROW_BYTES;              // [entry][frame][bone]
constexpr int G_BYTES = 2 * G_PIECE_BYTES + 64;                   // + what the last row's k-slots 24..31 read
constexpr size_t LDS_BYTES = (size_t)A_BYTES + G_BYTES + (size_t)TR_FLOATS * sizeof(float) + 64;
constexpr int SKIN_TILE_BYTES = 2 * 2 * 1024;                     // [k-step][piece][lane][8 bf16]
static_assert(SKIN_TILE_BYTES == MESH_SKIN_BF16_TILE_BYTES, "api.hip packs what this kernel reads");
static_assert(NB <= 24, "the bones of a row fit its 24 slots");
static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
}  // namespace ms

__global__ __launch_bounds__(mb::NW * 64) void mesh_rows_bf16s_kernel(MeshSkinArgs a) {
  X3_EXCLUSIVE_SIMD();
  using namespace ms;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  unsigned short* Ab = reinterpret_cast<unsigned short*>(lds);
  char* Gb = reinterpret_cast<char*>(lds) + A_BYTES;
  float* TRs = reinterpret_cast<float*>(Gb + G_BYTES);
  const int T = a.T, V = a.V;
  const int f0 = blockIdx.x * BM;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- staging: feature pieces (as mesh_rows_bf16_kernel), bone transforms as bf16 pieces, translations
  {
    const float* __restrict__ feat = a.feat;
    for (int i = tid; i < BM * LDA; i += NW * 64) {
      const int r = i / LDA, c = i - r * LDA;
      const int row = f0 + r < T ? f0 + r : T - 1;
      const int src = c < M1 ? c : (c < M2 ? c : c - (M2 - M1));
      const float x = c < KCOLS ? feat[(size_t)row * 200 + src] : 0.f;
      const unsigned short p0 = bf16_rne(x);
      const float r1 = x - bf16_f32(p0);
      const unsigned short p1 = bf16_rne(r1);
      const unsigned short p2 = bf16_rne(r1 - bf16_f32(p1));
      Ab[i] = c < M2 ? p0 : p1;
      Ab[A_PIECE_BYTES / 2 + i] = c < M1 ? p1 : (c < M2 ? p2 : p0);
    }
    const float* __restrict__ xf = a.xf;
    for (int i = tid; i < BM * NB * 3; i += NW * 64) {   // (frame r, bone b, transform row q): four entries of one row
      const int r = i / (NB * 3), bq = i - r * (NB * 3), b = bq / 3, q = bq - b * 3;
      const int row = f0 + r < T ? f0 + r : T - 1;
      const f32x4 g = *reinterpret_cast<const f32x4*>(xf + ((size_t)row * NB * 3 + bq) * 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned short hi = bf16_rne(g[k]);
        const unsigned short lo = bf16_rne(g[k] - bf16_f32(hi));
        char* dst = Gb + ((q * 4 + k) * BM + r) * G_ROW_BYTES + b * 2;
        *reinterpret_cast<unsigned short*>(dst) = hi;
        *reinterpret_cast<unsigned short*>(dst + G_PIECE_BYTES) = lo;
      }
    }
    for (int i = tid; i < 2 * 12 * BM; i += NW * 64)     // the unused bone slots 22, 23 of every row
      *reinterpret_cast<unsigned*>(Gb + i * G_ROW_BYTES + NB * 2) = 0u;
    if (tid < 16) *reinterpret_cast<unsigned*>(Gb + 2 * G_PIECE_BYTES + tid * 4) = 0u;
    if (tid < BM) {
      const int row = f0 + tid < T ? f0 + tid : T - 1;
      f32x4 t{0.f, 0.f, 0.f, 0.f};
      if (a.trans) { t[0] = a.trans[(size_t)row * 3]; t[1] = a.trans[(size_t)row * 3 + 1]; t[2] = a.trans[(size_t)row * 3 + 2]; }
      *reinterpret_cast<f32x4*>(TRs + tid * 4) = t;
    }
  }
  __syncthreads();

  const int n_tiles = (V + 31) / 32;
  const int per_block = (n_tiles + gridDim.y - 1) / gridDim.y;
  const int first = blockIdx.y * per_block;
  const int end = min(first + per_block, n_tiles);
  int vt = first + wave;
  if (vt >= end) return;

  const __amdgpu_buffer_rsrc_t wtab = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(a.wc_bf16), 0, (int)((size_t)n_tiles * TILE_BYTES), 0x00020000);
  const __amdgpu_buffer_rsrc_t stab = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(a.skin_bf16), 0, (int)((size_t)n_tiles * SKIN_TILE_BYTES), 0x00020000);
  const int lane16 = lane * 16;
  auto wload = [&](int tile_off, int ks, int c, int p) {
    const int f = c * 2 + p;
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wtab, lane16 + (f & 3) * 1024,
                                                                           tile_off + ks * 6144 + (f >> 2) * 4096, 0));
  };
  auto bload = [&](f32x4 (&dst)[3][2], int tile_off, int ks) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int p = 0; p < 2; ++p) dst[c][p] = wload(tile_off, ks, c, p);
  };
  auto tile_off_of = [&](int t) { return __builtin_amdgcn_readfirstlane(t) * TILE_BYTES; };
  const char* a_lane = reinterpret_cast<const char*>(Ab) + l31 * (LDA * 2) + lh * 16;
  auto aload = [&](f32x4 (&dst)[2][2], int ks) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int p = 0; p < 2; ++p)
        dst[i][p] = *reinterpret_cast<const f32x4*>(a_lane + p * A_PIECE_BYTES + i * 32 * (LDA * 2) + ks * 32);
  };
  const char* g_lane = Gb + l31 * G_ROW_BYTES + lh * 16;   // + (entry * 64 + half * 32) rows, + k-step * 32, + piece
  epi_gbyte_t vbase = (epi_gbyte_t)a.vertices;
  const size_t vrow_bytes = (size_t)V * 12;

  f32x4 ring[RING][3][2];
  f32x4 fa[2][2][2];
  {
    const int b0 = tile_off_of(vt);
#pragma unroll
    for (int ks = 0; ks < RING - 1; ++ks) bload(ring[ks], b0, ks);
  }
  aload(fa[0], 0);

#pragma unroll 1
  for (; vt < end; vt += NW) {
    const int bt = tile_off_of(vt);
    const int bnext = vt + NW < end ? tile_off_of(vt + NW) : bt;
    const int s = vt * 32 + l31;
    // this tile's skin-weight fragments [k-step][piece]: requested now, needed after the K loop
    f32x4 wf[2][2];
    {
      const int so = __builtin_amdgcn_readfirstlane(vt) * SKIN_TILE_BYTES;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int p = 0; p < 2; ++p)
          wf[ks][p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(stab, lane16 + (ks * 2 + p) * 1024, so, 0));
    }
    f32x16 acc[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][c][r] = 0.f;

#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {   // (the K loop of mesh_rows_bf16_kernel)
      const int kn = ks + RING - 1;
      f32x4 (&bn)[3][2] = ring[kn % RING];
      f32x4 (&fn)[2][2] = fa[(ks + 1) & 1];
      const int an = ks + 1 < KS ? ks + 1 : 0;
      const f32x4 (&fc)[2][2] = fa[ks & 1];
      const f32x4 (&bc)[3][2] = ring[ks % RING];
#pragma unroll
      for (int prod = 0; prod < 3; ++prod) {
        if (kn < KS) {
#pragma unroll
          for (int p = 0; p < 2; ++p) bn[prod][p] = wload(bt, kn, prod, p);
        }
        if (prod < 2) {
#pragma unroll
          for (int p = 0; p < 2; ++p)
            fn[prod][p] = *reinterpret_cast<const f32x4*>(a_lane + p * A_PIECE_BYTES + prod * 32 * (LDA * 2) + an * 32);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const bf16x8 av = __builtin_bit_cast(bf16x8, fc[i][prod == 0 ? 1 : 0]);
            const bf16x8 bv = __builtin_bit_cast(bf16x8, bc[c][prod == 1 ? 1 : 0]);
            acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[i][c], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0x6);
      }
    }
#pragma unroll
    for (int ks = 0; ks < RING - 1; ++ks) bload(ring[ks], bnext, ks);

    // ---- bone blend on the matrix cores + out = T^R v + T^t on registers
    const char* trl = reinterpret_cast<const char*>(TRs) + lh * 64;
    const unsigned lane_off = ((unsigned)(f0 + 4 * lh) * (unsigned)V + (unsigned)s) * 12u;
    const bool full = f0 + BM <= T;
    // Six stages (frame half i, transform row q), software-pipelined: the 24 MFMAs of stage st + 1 are issued between the
    // vector instructions that apply stage st (a wave's MFMAs run in the matrix pipe while it issues vector work; in
    // program order "24 MFMAs, then the apply" the two phases simply followed each other and the kernel was SLOWER than
    // vector skinning: 14.7 against 16.9 M frames/s).  Order pinned with full scheduling fences; consecutive MFMAs always
    // go to different accumulators.
    f32x16 tk[2][4];
    f32x4 ga[2][4][2];   // [k-step][entry][piece] of the stage being issued
    auto gload = [&](int st) {
      const int i = st / 3, q = st - i * 3;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int p = 0; p < 2; ++p)
            ga[ks][k][p] = *reinterpret_cast<const f32x4*>(g_lane + p * G_PIECE_BYTES +
                                                           ((q * 4 + k) * BM + i * 32) * G_ROW_BYTES + ks * 32);
    };
    auto zero = [&](f32x16 (&t)[4], int st) {   // the translation entry starts from the frame's global translation:
      const int i = st / 3, q = st - i * 3;     // sixteen LDS reads in one go here instead of one (and its latency) per frame
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) t[k][r] = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        t[3][r] = *reinterpret_cast<const float*>(trl + (i * 32 + (r & 3) + 8 * (r >> 2)) * 16 + q * 4);
    };
    auto mma = [&](f32x16 (&t)[4], int g) {   // MFMA g of a stage's 24: product g / 8, k-step (g / 4) % 2, entry g % 4
      const int prod = g >> 3, ks = (g >> 2) & 1, k = g & 3;
      const bf16x8 av = __builtin_bit_cast(bf16x8, ga[ks][k][prod == 0 ? 1 : 0]);
      const bf16x8 bv = __builtin_bit_cast(bf16x8, wf[ks][prod == 1 ? 1 : 0]);
      t[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, t[k], 0, 0, 0);
    };
    auto apply = [&](int st, const f32x16 (&t)[4], int r) {
      const int i = st / 3, q = st - i * 3;
      const int dm = i * 32 + (r & 3) + 8 * (r >> 2);
      const float o = __builtin_fmaf(t[0][r], acc[i][0][r],
                                     __builtin_fmaf(t[1][r], acc[i][1][r], __builtin_fmaf(t[2][r], acc[i][2][r], t[3][r])));
      if (s < V && (full || f0 + 4 * lh + dm < T))
        *(epi_gfloat_t)(vbase + (size_t)dm * vrow_bytes + (lane_off + 4u * q)) = o;
    };
    gload(0);
    zero(tk[0], 0);
#pragma unroll
    for (int g = 0; g < 24; ++g) mma(tk[0], g);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int st = 0; st < 6; ++st) {
      if (st + 1 < 6) {
        gload(st + 1);
        zero(tk[(st + 1) & 1], st + 1);
      }
#pragma unroll
      for (int g = 0; g < 24; ++g) {   // one MFMA of the next stage, then the vector work of one frame of this one
        if (st + 1 < 6) mma(tk[(st + 1) & 1], g);
        if (g < 16) apply(st, tk[st & 1], g);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
}

hipError_t launch_mesh_rows_bf16s(const MeshSkinArgs& a, hipStream_t stream) {
  if (!a.skin_bf16 || !a.wc_bf16) return hipErrorInvalidValue;
  if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(mesh_rows_bf16s_kernel), ms::LDS_BYTES)) return e;
  const int bx = (a.T + mb::BM - 1) / mb::BM;
  const int n_tiles = (a.V + 31) / 32;
  int by = bx >= 256 ? 1 : (256 + bx - 1) / bx;
  const int max_by = (n_tiles + mb::NW - 1) / mb::NW;
  if (by > max_by) by = max_by;
  hipLaunchKernelGGL(mesh_rows_bf16s_kernel, dim3(bx, by), dim3(mb::NW * 64), ms::LDS_BYTES, stream, a);
  return hipGetLastError();
}

hipError_t launch_mesh_rows_bf16(const MeshSkinArgs& a, hipStream_t stream) {
  if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(mesh_rows_bf16_kernel<false>), mb::LDS_BYTES)) return e;
  if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(mesh_rows_bf16_kernel<true>), mb::LDS_BYTES)) return e;
  const int bx = (a.T + mb::BM - 1) / mb::BM;
  const int n_tiles = (a.V + 31) / 32;
  int by = bx >= 256 ? 1 : (256 + bx - 1) / bx;
  const int max_by = (n_tiles + mb::NW - 1) / mb::NW;
  if (by > max_by) by = max_by;
  if (a.kb > 4)
    hipLaunchKernelGGL(mesh_rows_bf16_kernel<true>, dim3(bx, by), dim3(mb::NW * 64), mb::LDS_BYTES, stream, a);
  else
    hipLaunchKernelGGL(mesh_rows_bf16_kernel<false>, dim3(bx, by), dim3(mb::NW * 64), mb::LDS_BYTES, stream, a);
  return hipGetLastError();
}

hipError_t launch_mesh_rows(const MeshSkinArgs& a, hipStream_t stream) {
  if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(mesh_rows_kernel<false>), mr::LDS_BYTES)) return e;
  if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(mesh_rows_kernel<true>), mr::LDS_BYTES)) return e;
  const int bx = (a.T + mr::BM - 1) / mr::BM;
  const int n_tiles = (a.V + 31) / 32;
  // fewer than one workgroup per CU: split the mesh's tiles over grid.y (at least one tile per wave)
  int by = bx >= 256 ? 1 : (256 + bx - 1) / bx;
  const int max_by = (n_tiles + mr::NW - 1) / mr::NW;
  if (by > max_by) by = max_by;
  if (a.kb > 4)
    hipLaunchKernelGGL(mesh_rows_kernel<true>, dim3(bx, by), dim3(mr::NW * 64), mr::LDS_BYTES, stream, a);
  else
    hipLaunchKernelGGL(mesh_rows_kernel<false>, dim3(bx, by), dim3(mr::NW * 64), mr::LDS_BYTES, stream, a);
  return hipGetLastError();
}

}  // namespace empose
