// Full-mesh SMPL-H evaluation (mesh.hip: v_posed = wc . [pose features, shape, 1], then linear blend skinning of all V
// vertices; reference bodymodels/smpl.py:81-147) with the blend-shape contraction on the bf16 matrix cores in THREE bf16
// pieces per operand -- x = x0 + x1 + x2, every piece the round-to-nearest bf16 of what the previous ones leave, six
// v_mfma_f32_32x32x16_bf16 products per 16 k (x0w0, x0w1, x1w0, x0w2, x1w1, x2w0, the small ones first), fp32 accumulate:
// the fp32-EQUIVALENT arithmetic of mlp_fused_x3.hip / lstm_x3.hip (bf16x3.h), on ALL 200 columns.  (mesh.hip's
// mesh_rows_bf16_kernel keeps two pieces / three products on the 189 pose columns: not fp32-equivalent, opt-in.)
//
// Blocking as mesh_rows_kernel: a workgroup owns 64 frames; their features -- split once into three piece planes, 81 KB --
// and their 22 relative bone transforms (66 KB, fp32) stay in LDS; its four waves (ONE per SIMD: bf16_hazard_repro.md) walk
// the 32-vertex tiles.  A tile is 13 k-steps x 6 products x 6 accumulators (2 frame tiles x 3 coordinate planes) = 468
// MFMAs; the coefficients arrive from L2 in fragment order ([tile][k-step][plane][piece] -> 1 KB, api.hip
// pack_mesh_tiles_x3; 117 KB per tile) through a register ring RING - 1 k-steps ahead.
//
// The skinning of a tile (4 bones x 3 rows x 16 bytes of LDS per (frame, vertex): 393 KB per tile and wave, LDS-bound,
// ~12 k clocks) is SOFTWARE-PIPELINED under the next tile's K loop (template OVERLAP): two accumulator sets, the vector
// work of tile n placed between the MFMA groups of tile n + 1 -- with one wave per SIMD nothing else can hide it.
#include <hip/hip_runtime.h>
#include <type_traits>

#include "bf16x3.h"
#include "gemm_epilogue.h"
#include "kernels.h"

namespace empose {

typedef float mx_f32x2 __attribute__((ext_vector_type(2)));

namespace mx {
constexpr int BM = 64, NW = 4;
constexpr int K = 200, KS = 13;                 // 200 columns -> 13 k-steps of 16 (columns 200..207 are zero)
constexpr int LDA = 216;                        // bf16 per row of a piece plane: 432 B = 27 x 16 (odd: conflict-free b128 reads)
constexpr int A_PIECE_BYTES = BM * LDA * 2;     // 27,648
constexpr int A_BYTES = 3 * A_PIECE_BYTES;      // 82,944
constexpr int XF_FLOATS = BM * NB * 12;
constexpr int TR_FLOATS = BM * 4;
constexpr size_t LDS_BYTES = (size_t)A_BYTES + (size_t)(XF_FLOATS + TR_FLOATS) * sizeof(float) + 64;
constexpr int TILE_BYTES = KS * 9 * 1024;       // packed coefficients of one 32-vertex tile: [k-step][plane][piece]
constexpr int RING = 3;                         // B fragments: steps s + 1, s + 2 in flight while step s is consumed
static_assert(TILE_BYTES == MESH_X3_TILE_BYTES, "api.hip packs what this kernel reads");
static_assert(LDS_BYTES <= 160 * 1024 && LDS_BYTES > 80 * 1024, "one workgroup per CU");
}  // namespace mx

template <bool OVERLAP>
__global__ __launch_bounds__(mx::NW * 64) void mesh_rows_x3_kernel(MeshSkinArgs a) {
  X3_EXCLUSIVE_SIMD();
  using namespace mx;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  unsigned short* Ab = reinterpret_cast<unsigned short*>(lds);
  float* XFs = reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + A_BYTES);
  float* TRs = XFs + XF_FLOATS;
  const int T = a.T, V = a.V;
  const int f0 = blockIdx.x * BM;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- staging: the features as three bf16 piece planes, relative transforms, translations (rows past T repeat row T - 1)
  {
    const float* __restrict__ feat = a.feat;
    for (int i = tid; i < BM * (LDA / 2); i += NW * 64) {
      const int r = i / (LDA / 2), c = (i - r * (LDA / 2)) * 2;
      const int row = f0 + r < T ? f0 + r : T - 1;
      const float x0 = c < K ? feat[(size_t)row * K + c] : 0.f, x1 = c + 1 < K ? feat[(size_t)row * K + c + 1] : 0.f;
      unsigned h, m, l;
      split_pair(x0, x1, h, m, l);
      unsigned* dst = reinterpret_cast<unsigned*>(Ab) + r * (LDA / 2) + c / 2;
      dst[0] = h; dst[A_PIECE_BYTES / 4] = m; dst[2 * (A_PIECE_BYTES / 4)] = l;
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

  // The coefficient table as a raw buffer (mesh.hip): lane offset in a VGPR, tile / k-step offset in an SGPR.
  const __amdgpu_buffer_rsrc_t wtab = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(a.wc_x3), 0, (int)((size_t)n_tiles * TILE_BYTES), 0x00020000);
  const int lane16 = lane * 16;
  auto wload = [&](int tile_off, int ks, int f) {    // fragment f = plane * 3 + piece of k-step ks
    return __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(wtab, lane16 + (f & 3) * 1024,
                                                                             tile_off + ks * 9216 + (f >> 2) * 4096, 0));
  };
  auto tile_off_of = [&](int t) { return __builtin_amdgcn_readfirstlane(t) * TILE_BYTES; };
  // A fragments of this lane: row l31 (+ 32), 16 bytes at k-step * 32 + lh * 16; [frame tile][piece]
  const char* a_lane = reinterpret_cast<const char*>(Ab) + l31 * (LDA * 2) + lh * 16;
  auto aread = [&](int i, int p, int ks) {
    return *reinterpret_cast<const u32x4_t*>(a_lane + p * A_PIECE_BYTES + i * 32 * (LDA * 2) + ks * 32);
  };
  epi_gbyte_t vbase = (epi_gbyte_t)a.vertices;
  const size_t vrow_bytes = (size_t)V * 12;
  const char* xfl = reinterpret_cast<const char*>(XFs) + lh * (4 * NB * 48);
  const char* trl = reinterpret_cast<const char*>(TRs) + lh * 64;
  const bool full = f0 + BM <= T;

  u32x4_t ring[RING][9];       // [slot][plane * 3 + piece]
  u32x4_t fa[2][2][3];         // [slot][frame tile][piece]
  {
    const int b0 = tile_off_of(vt);
#pragma unroll
    for (int ks = 0; ks < RING - 1; ++ks)
#pragma unroll
      for (int f = 0; f < 9; ++f) ring[ks][f] = wload(b0, ks, f);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int p = 0; p < 3; ++p) fa[0][i][p] = aread(i, p, 0);

  // What the skinning of a tile needs besides its accumulators.
  struct SkinTile { int s; int4 bone4; f32x4 w4; };
  auto skin_params = [&](int t) {
    SkinTile k;
    k.s = t * 32 + l31;
    k.bone4 = *reinterpret_cast<const int4*>(a.skin_idx4 + (size_t)k.s * 4);
    k.w4 = *reinterpret_cast<const f32x4*>(a.skin_w4 + (size_t)k.s * 4);
    return k;
  };
  // One (frame, vertex) pair e = frame tile * 16 + accumulator index: blended 3 x 4 transform of the vertex's four bones
  // (reference order: T = sum_k w_k G_k, v = T . [v_posed, 1] + trans), one mat-vec, one 12-byte store.
  float lab_sink = 0.f;   // (lab builds only: what a dummy skinning pass leaves)
  // The result goes to registers (outv): nothing is stored here.  A global store issued while this
  // SIMD has v_mfma_f32_32x32x16_bf16 in flight corrupts one accumulator element of the running products -- element r = 1,
  // lanes 48..63 of the first accumulator of a group, the very footprint scripts/dev/bf16_hazard_repro.md records for two
  // waves per SIMD (there the OTHER wave's skinning stores met this wave's MFMAs); found with scripts/dev/mesh_x3_lab.sh:
  // the same vector work and LDS reads between the MFMAs WITHOUT the stores is clean.  So the vertices of a tile leave in
  // `flush`, after the pass, when the wave has no MFMA in flight.
  float outv[32][3];      // the skinned vertices of the tile being skinned, until `flush`
  auto skin_one = [&](const f32x16 (&acc)[2][3], const SkinTile& st, int e, bool dummy = false) {
    const int i = e >> 4, r = e & 15;
    const int dm = i * 32 + (r & 3) + 8 * (r >> 2);   // frame within the block, less 4 * lh
    const float vx = acc[i][0][r], vy = acc[i][1][r], vz = acc[i][2][r];
#if defined(MX_LAB_DUMMY) && defined(MX_LAB_NOLDS)
    const f32x4 tr = dummy ? f32x4{vx, vy, vz, vx} : *reinterpret_cast<const f32x4*>(trl + dm * 16);
#else
    const f32x4 tr = *reinterpret_cast<const f32x4*>(trl + dm * 16);
#endif
    const char* xk[4] = {xfl + st.bone4.x * 48, xfl + st.bone4.y * 48, xfl + st.bone4.z * 48, xfl + st.bone4.w * 48};
    float out[3];
#pragma unroll
    for (int row = 0; row < 3; ++row) {
#if defined(MX_LAB_DUMMY) && defined(MX_LAB_NOLDS)
#define MX_GK(k) (dummy ? f32x4{vx + (float)(k), vy, vz, st.w4[k]} : *reinterpret_cast<const f32x4*>(xk[k] + dm * (NB * 48) + row * 16))
#else
#define MX_GK(k) (*reinterpret_cast<const f32x4*>(xk[k] + dm * (NB * 48) + row * 16))
#endif
      f32x4 gk = MX_GK(0);
      mx_f32x2 Ta = mx_f32x2{st.w4[0], st.w4[0]} * mx_f32x2{gk[0], gk[1]}, Tb = mx_f32x2{st.w4[0], st.w4[0]} * mx_f32x2{gk[2], gk[3]};
#pragma unroll
      for (int k = 1; k < 4; ++k) {
        gk = MX_GK(k);
        Ta = __builtin_elementwise_fma(mx_f32x2{st.w4[k], st.w4[k]}, mx_f32x2{gk[0], gk[1]}, Ta);
        Tb = __builtin_elementwise_fma(mx_f32x2{st.w4[k], st.w4[k]}, mx_f32x2{gk[2], gk[3]}, Tb);
      }
      out[row] = __builtin_fmaf(Ta[0], vx, __builtin_fmaf(Ta[1], vy, __builtin_fmaf(Tb[0], vz, Tb[1]))) + tr[row];
    }
#ifdef MX_LAB_DUMMY
    if (dummy) { lab_sink += out[0] + out[1] + out[2]; return; }
#endif
    // (pinned: with the stores gone nothing anchors an element -- the compiler sank the arithmetic of all 32 of them down
    // to the flush and kept their 384 fetched transform rows alive until then, 1500 spilled registers; an empty asm that
    // "uses" the three results and clobbers memory holds both the arithmetic and the LDS reads of an element in place)
    asm volatile("" : "+v"(out[0]), "+v"(out[1]), "+v"(out[2]) : : "memory");
    outv[e][0] = out[0]; outv[e][1] = out[1]; outv[e][2] = out[2];
  };
  // The skinned vertices of a tile -> memory: 32 stores of 12 bytes per lane, as buffer stores on the block's 64 output
  // rows (lane offset in ONE VGPR, the frame's row offset in an SGPR: no 64-bit address pair per element -- with those the
  // 96 values waiting for the flush pushed the kernel into 1900 spills).
  const __amdgpu_buffer_rsrc_t vout = __builtin_amdgcn_make_buffer_rsrc(
      a.vertices + (size_t)f0 * V * 3, 0, (int)((size_t)min(BM, T - f0) * vrow_bytes), 0x00020000);
  auto flush = [&](auto check_tag, const SkinTile& st) {
    constexpr bool CHECK = decltype(check_tag)::value;   // CHECK: the mesh's last vertices may not exist (rows past T are
                                                         // outside the buffer: the hardware drops those stores)
    const int lane_off = (4 * lh * V + st.s) * 12;
    if (CHECK && st.s >= V) return;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dm = i * 32 + (r & 3) + 8 * (r >> 2);
        const int e = i * 16 + r;
        typedef float mx_f32x3 __attribute__((ext_vector_type(3)));
        __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(__attribute__((ext_vector_type(3))) unsigned,
                                                                 mx_f32x3{outv[e][0], outv[e][1], outv[e][2]}),
                                              vout, lane_off, dm * (int)vrow_bytes, 0);
      }
  };

  // One pass: the K loop of tile `bt` into acc_k (DO_K) with the skinning of the previous tile (acc_s, st) between its MFMA
  // groups (DO_S).  78 groups of six MFMAs on six different accumulators; 32 skinning elements; the coefficient prefetch
  // (nine fragments per k-step) and the A-fragment reads (six) ride in the first groups of a k-step.
  // `flush_at` (DO_K without DO_S): the k-step in front of which the vertices of the PREVIOUS tile (outv, st_flush) are
  // stored, -1 for none -- see the tile loop.
  auto pass = [&](auto do_k_tag, auto do_s_tag, f32x16 (&acc_k)[2][3], const f32x16 (&acc_s)[2][3],
                  const SkinTile& st, int bt, int bnext, bool dummy = false, int flush_at = -1,
                  const SkinTile* st_flush = nullptr) {
    constexpr bool DO_K = decltype(do_k_tag)::value, DO_S = decltype(do_s_tag)::value;
    if (DO_K) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc_k[i][c][r] = 0.f;
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int kn = ks + RING - 1;
      u32x4_t (&bn)[9] = ring[kn % RING];
      u32x4_t (&fn)[2][3] = fa[(ks + 1) & 1];
      const int an = ks + 1 < KS ? ks + 1 : 0;
      const u32x4_t (&fc)[2][3] = fa[ks & 1];
      const u32x4_t (&bc)[9] = ring[ks % RING];
      if (DO_K && !DO_S && ks == flush_at) {
        // no MFMA of this wave may be in flight when a store issues (see skin_one): let the last group of the step before
        // drain (six MFMAs of 8 passes: 4 x 16 idle cycles cover the tail), store, and only then go on
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
        flush(std::false_type{}, *st_flush);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int prod = 0; prod < 6; ++prod) {
        const int g = ks * 6 + prod;
        if (DO_K) {
          // this group's share of the prefetch: coefficients of step ks + 2 (of the next tile past this one's last steps:
          // every tile starts in ring slot 0 because KS % RING == 1 -- see the slot arithmetic below), A fragments of ks + 1
          if (prod < 3) {
#pragma unroll
            for (int q = 0; q < 3; ++q)
              bn[prod * 3 + q] = kn < KS ? wload(bt, kn, prod * 3 + q) : wload(bnext, kn - KS, prod * 3 + q);
          }
          fn[prod & 1][prod >> 1] = aread(prod & 1, prod >> 1, an);
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int c = 0; c < 3; ++c)
              acc_k[i][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fc[i][X3_PA[prod]]),
                                                                    __builtin_bit_cast(bf16x8_t, bc[c * 3 + X3_PB[prod]]),
                                                                    acc_k[i][c], 0, 0, 0);
        }
        if (DO_S) {
#pragma unroll
          for (int e = (g * 32) / (KS * 6); e < ((g + 1) * 32) / (KS * 6); ++e) skin_one(acc_s, st, e, dummy);
        }
        if (DO_K) __builtin_amdgcn_sched_barrier(0x6);   // only VALU / SALU may move across: the MFMA order stands
        else __builtin_amdgcn_sched_barrier(0);           // (skinning alone: keep the elements apart, or the scheduler
                                                          // hoists hundreds of LDS reads and spills)
      }
    }
    if (DO_K) {
      // The seam: KS = 13 is neither a multiple of the ring depth nor even, so the first two k-steps of the next tile (fetched
      // during steps 11 and 12) sit in slots 1 and 2 and its first A fragments in slot 1 -- moved to where step 0 looks for
      // them (96 register moves per tile, 1 % of its clocks; the loads were issued two k-steps ago).
#pragma unroll
      for (int f = 0; f < 9; ++f) { ring[0][f] = ring[KS % RING][f]; ring[1][f] = ring[(KS + 1) % RING][f]; }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) fa[0][i][p] = fa[KS & 1][i][p];
    }
  };

  f32x16 accA[2][3], accB[2][3];
  // Whole tiles of existing frames take the unchecked flush; the mesh's last, partial tile (V % 32 vertices) and the last
  // workgroup of a launch whose T is not a multiple of 64 take the checked one, one after the other.
  const int end_fast = full ? min(end, V / 32) : first;
  const std::true_type yes{};
  const std::false_type no{};
  if (vt < end_fast) {
    if (OVERLAP) {
      // tile 0: K loop only; tiles 1 ..: K loop + skinning of the one before; then the last tile's skinning
      SkinTile st = skin_params(vt);
      {
        const int bt = tile_off_of(vt);
        const int bnext = vt + NW < end ? tile_off_of(vt + NW) : bt;
        pass(yes, no, accA, accA, st, bt, bnext);
      }
#pragma unroll 1
      for (vt += NW; vt < end_fast; vt += NW) {
        const int bt = tile_off_of(vt);
        const int bnext = vt + NW < end ? tile_off_of(vt + NW) : bt;
        const SkinTile st_next = skin_params(vt);
#ifdef MX_LAB_DUMMY   // (lab: the real skinning one after the other, a DUMMY one -- results dropped -- under the K loop)
        pass(yes, yes, accB, accA, st, bt, bnext, true);
        pass(no, yes, accB, accA, st, 0, 0);
#else
        pass(yes, yes, accB, accA, st, bt, bnext);
#endif
#ifndef MX_LAB_NOFLUSH
        flush(no, st);
#endif
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int c = 0; c < 3; ++c) accA[i][c] = accB[i][c];
        st = st_next;
      }
      pass(no, yes, accB, accA, st, 0, 0);
      flush(no, st);
#ifdef MX_LAB_DUMMY
      if (lab_sink == 12345.678f) a.vertices[0] = lab_sink;
#endif
    } else {
      // K loop, skinning, stores, one after the other -- but the stores of a tile wait until the NEXT tile's K loop has
      // reached k-step `flush_at`, a different one for every wave and workgroup (a.stagger): every wave of the chip
      // finishing its tile's 24 KB at the same moment made the stores a phase of their own (0.27 ms of 1.30 at 16384
      // frames, the HBM write time of the 1.35 GB); spread over the tile's period they disappear under the products.
      const int flush_at = a.stagger ? (1 + 3 * wave + (int)(blockIdx.x % 3)) : -1;
      bool pending = false;
      SkinTile st_prev = skin_params(vt);
#pragma unroll 1
      for (; vt < end_fast; vt += NW) {
        const int bt = tile_off_of(vt);
        const int bnext = vt + NW < end ? tile_off_of(vt + NW) : bt;
        const SkinTile st = skin_params(vt);
#ifndef MX_LAB_NOK        // (lab, scripts/dev/mesh_x3_lab.sh: what each phase of a tile costs; results are then wrong)
        pass(yes, no, accA, accA, st, bt, bnext, false, pending ? flush_at : -1, &st_prev);
#endif
#ifndef MX_LAB_NOSKIN
        pass(no, yes, accB, accA, st, 0, 0);
#endif
#ifndef MX_LAB_NOFLUSH
        if (flush_at < 0) flush(no, st);
        else { st_prev = st; pending = true; }
#endif
      }
      if (pending) flush(no, st_prev);
    }
  }
#pragma unroll 1
  for (; vt < end; vt += NW) {
    const int bt = tile_off_of(vt);
    const int bnext = vt + NW < end ? tile_off_of(vt + NW) : bt;
    const SkinTile st = skin_params(vt);
    pass(yes, no, accA, accA, st, bt, bnext);
    pass(no, yes, accB, accA, st, 0, 0);
    flush(yes, st);
  }
}

hipError_t launch_mesh_rows_x3(const MeshSkinArgs& a, bool overlap, hipStream_t stream) {   // (a.stagger: see the tile loop)
  if (!a.wc_x3 || a.kb > 4) return hipErrorInvalidValue;
  const int bx = (a.T + mx::BM - 1) / mx::BM;
  const int n_tiles = (a.V + 31) / 32;
  int by = bx >= 256 ? 1 : (256 + bx - 1) / bx;
  const int max_by = (n_tiles + mx::NW - 1) / mx::NW;
  if (by > max_by) by = max_by;
  if (overlap) {
    if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(mesh_rows_x3_kernel<true>), mx::LDS_BYTES)) return e;
    hipLaunchKernelGGL(mesh_rows_x3_kernel<true>, dim3(bx, by), dim3(mx::NW * 64), mx::LDS_BYTES, stream, a);
  } else {
    if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(mesh_rows_x3_kernel<false>), mx::LDS_BYTES)) return e;
    hipLaunchKernelGGL(mesh_rows_x3_kernel<false>, dim3(bx, by), dim3(mx::NW * 64), mx::LDS_BYTES, stream, a);
  }
  return hipGetLastError();
}

}  // namespace empose
