// SMPL-H on the sensor sub-mesh: everything of one body-model evaluation that is not a matrix product.
//
//   update_feat    theta/beta update of the LGD step (reference models.py:588-592), Rodrigues per joint
//                  (angle guard switchable: smplx ||r + 1e-8|| or the so3 clamp), GEMM feature row [vec(R_j - I) | beta | 1]
//   chain_sensors  22-joint kinematic chain, linear blend skinning of the ~84 needed vertices, vertex normals and
//                  sensor frames (reference virtual_sensors.py:16-38, utils.py:126-146), sensor offsets
//                  (reference models.py:478-479), the reconstruction residual (reference loss.py:23-41) and its
//                  hand-derived reverse pass down to d v_posed, d J and d R (oracle/analytic_np.py is the blueprint)
//   rodrigues_bwd  d R -> d theta, and the gradient features written into the network input row
//
// chain_sensors runs a small tile of frames per workgroup; all per-frame state (rotations, rest joints, global
// transforms, skinned vertices and their cotangents, per-bone force/moment sums) lives in LDS and every phase is a
// flat parallel-for over (frame, item) so that no lane idles on the serial structure of the skeleton: the forward
// chain is a per-(joint,row) walk down the root path, the reverse chain uses the world-space form
// dR_j = G_p^T (sum_subtree M_b - F_b (x) t_j) G_j, which needs subtree sums instead of a level-by-level sweep.
#include "kernels.h"
#include "smpl_math.h"
#include "feat_rows.h"

#include <cstdint>
#include <cstdlib>

namespace empose {


// ---------------------------------------------------------------------------------------------------------------
__global__ void pack_inputs_kernel(PackArgs a) {
  const int T = a.B * a.F;
  const int d_in = a.n_markers * 12;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= T * (d_in + 1)) return;
  const int t = idx / (d_in + 1), c = idx % (d_in + 1);
  if (c == d_in) {
    if (!a.frame_scale) return;
    const int b = t / a.F, f = t % a.F;
    const int len = a.seq_lengths ? a.seq_lengths[b] : a.F;
    // reference: loss / n_frames, then grad * B * F  ->  F / len per valid frame (models.py:578-579, loss.py:36-39);
    // unpadded mode: every row is treated as a window of its own length (F == len), i.e. weight 1
    float s = (f < len) ? (a.rows_as_unpadded ? 1.f : (float)a.F / (float)len) : 0.f;
    if (a.marker_masks) {
      bool all = true;
      for (int m = 0; m < 12; ++m) all = all && (a.marker_masks[(size_t)t * 12 + m] != 0.f);
      if (!all) s = 0.f;
    }
    a.frame_scale[t] = s;
    return;
  }
  float v;
  int marker;
  if (c < a.n_markers * 3) {
    const int mi = c / 3, k = c % 3;
    marker = a.marker_idx[mi];
    v = a.marker_pos[(size_t)t * 36 + marker * 3 + k];
  } else {
    const int cc = c - a.n_markers * 3;
    const int mi = cc / 9, k = cc % 9;
    marker = a.marker_idx[mi];
    v = a.marker_oris[(size_t)t * 108 + marker * 9 + k];
  }
  if (a.suppress_missing && a.marker_masks) {   // the reference's arithmetic, so that a NaN reading stays a NaN
    const bool valid = a.marker_masks[(size_t)t * 12 + marker] == 1.f;
    v = v * (valid ? 1.f : 0.f) + (0.f + a.mask_value) * (valid ? 0.f : 1.f);
  }
  a.x[(size_t)t * a.ldx + c] = v;
}

// All twelve sensors in their own order and nothing to replace (the headline configuration): the input columns of a frame
// are its 36 position and 108 orientation values as they lie -- 36 pieces of 16 bytes per frame, no index arithmetic per
// element; thread 36 of a frame's group writes the frame weight.  (25.6 -> 13 us at 32768 frames.)
__global__ __launch_bounds__(256) void pack_inputs_all_kernel(PackArgs a) {
  const int T = a.B * a.F;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const int t = (int)(idx / 37), q = (int)(idx - (long)t * 37);
  if (t >= T) return;
  if (q == 36) {
    if (!a.frame_scale) return;
    const int b = t / a.F, f = t - b * a.F;
    const int len = a.seq_lengths ? a.seq_lengths[b] : a.F;
    float s = (f < len) ? (a.rows_as_unpadded ? 1.f : (float)a.F / (float)len) : 0.f;
    if (a.marker_masks) {
      bool all = true;
      for (int m = 0; m < 12; ++m) all = all && (a.marker_masks[(size_t)t * 12 + m] != 0.f);
      if (!all) s = 0.f;
    }
    a.frame_scale[t] = s;
    return;
  }
  typedef float f4 __attribute__((ext_vector_type(4)));
  const f4 v = q < 9 ? reinterpret_cast<const f4*>(a.marker_pos + (size_t)t * 36)[q]
                     : reinterpret_cast<const f4*>(a.marker_oris + (size_t)t * 108)[q - 9];
  reinterpret_cast<f4*>(a.x + (size_t)t * a.ldx)[q] = v;
}

hipError_t launch_pack_inputs(const PackArgs& a, hipStream_t stream) {
  bool plain = a.n_markers == 12 && !(a.suppress_missing && a.marker_masks) && a.ldx % 4 == 0 &&
               (((uintptr_t)a.marker_pos | (uintptr_t)a.marker_oris | (uintptr_t)a.x) & 15) == 0;
  for (int i = 0; i < 12; ++i) plain = plain && a.marker_idx[i] == i;
  if (plain) {
    const long n37 = (long)a.B * a.F * 37;
    hipLaunchKernelGGL(pack_inputs_all_kernel, dim3((unsigned)((n37 + 255) / 256)), dim3(256), 0, stream, a);
    return hipGetLastError();
  }
  const long n = (long)a.B * a.F * (a.n_markers * 12 + 1);
  hipLaunchKernelGGL(pack_inputs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a);
  return hipGetLastError();
}

// One thread per (frame, slot): slots 0..21 are joints, 22..31 the ten shape coefficients.  The two wide outputs (rot:
// 198 floats per frame, feat: 200) are assembled in LDS and written out by the whole block as contiguous 8 / 16-byte
// pieces; a lane writing its nine floats at a 36-byte stride reached 2.9 TB/s.
constexpr int UF_FRAMES = 8;   // frames per 256-thread block
__global__ __launch_bounds__(256) void update_feat_kernel(FeatArgs a) {
  __shared__ __attribute__((aligned(16))) float s_rot[UF_FRAMES * NB * 9];
  __shared__ __attribute__((aligned(16))) float s_feat[UF_FRAMES * 200];
  const int t0 = blockIdx.x * UF_FRAMES;
  const int fl = threadIdx.x >> 5, slot = threadIdx.x & 31;
  feat_frame(a, t0 + fl, slot, s_feat + fl * 200, s_rot + fl * NB * 9);   // feat_rows.h
  __syncthreads();
  // the block's frames are contiguous in both outputs
  const int nf = min(UF_FRAMES, a.T - t0);
  if (a.rot) {   // (the frame-per-lane kernel evaluates the rotations itself)
    const float2* src = reinterpret_cast<const float2*>(s_rot);
    float2* dst = reinterpret_cast<float2*>(a.rot + (size_t)t0 * NB * 9);   // 198 floats per frame: 8-byte pieces
    for (int i = threadIdx.x; i < nf * (NB * 9 / 2); i += 256) dst[i] = src[i];
  }
  {
    const float4* src = reinterpret_cast<const float4*>(s_feat);
    float4* dst = reinterpret_cast<float4*>(a.feat + (size_t)t0 * 200);
    for (int i = threadIdx.x; i < nf * 50; i += 256) dst[i] = src[i];
  }
}

hipError_t launch_update_feat(const FeatArgs& a, hipStream_t stream) {
  hipLaunchKernelGGL(update_feat_kernel, dim3((unsigned)((a.T + UF_FRAMES - 1) / UF_FRAMES)), dim3(256), 0, stream, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
__global__ void rodrigues_bwd_kernel(RodBwdArgs a) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int t = idx >> 5, slot = idx & 31;
  if (t >= a.T) return;
  if (slot < NB) {
    const float* th = a.theta + (size_t)t * a.ld_theta + slot * 3;
    Rod q; float R[9];
    rodrigues(th[0], th[1], th[2], a.rod_conv, q, R);
    float dR[9];
    const float* dr = a.d_rot + ((size_t)t * NB + slot) * 9;
#pragma unroll
    for (int e = 0; e < 9; ++e) dR[e] = dr[e];
    if (slot >= 1) {
      const float* df = a.d_feat + (size_t)t * 200 + (slot - 1) * 9;
#pragma unroll
      for (int e = 0; e < 9; ++e) dR[e] += df[e];
    }
    // K and K^2
    const float K[9] = {0.f, -q.dz, q.dy, q.dz, 0.f, -q.dx, -q.dy, q.dx, 0.f};
    const float KK[9] = {-q.dz * q.dz - q.dy * q.dy, q.dx * q.dy, q.dx * q.dz,
                         q.dx * q.dy, -q.dz * q.dz - q.dx * q.dx, q.dy * q.dz,
                         q.dx * q.dz, q.dy * q.dz, -q.dy * q.dy - q.dx * q.dx};
    float ds = 0.f, dc1 = 0.f;
#pragma unroll
    for (int e = 0; e < 9; ++e) { ds += dR[e] * K[e]; dc1 += dR[e] * KK[e]; }
    const float oc = 1.f - q.c;
    // dK = s dR + (1-c) (dR K^T + K^T dR)
    float dK[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float m = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) m += dR[r * 3 + k] * K[c * 3 + k] + K[k * 3 + r] * dR[k * 3 + c];
        dK[r * 3 + c] = q.s * dR[r * 3 + c] + oc * m;
      }
    const float ddx = dK[7] - dK[5], ddy = dK[2] - dK[6], ddz = dK[3] - dK[1];
    float da = ds * q.c + dc1 * q.s;
    da -= (ddx * q.dx + ddy * q.dy + ddz * q.dz) / q.ang;
    const float g0 = ddx / q.ang + da * q.ux / q.ang;
    const float g1 = ddy / q.ang + da * q.uy / q.ang;
    const float g2 = ddz / q.ang + da * q.uz / q.ang;
    float* g = a.g_theta + (size_t)t * a.ld_g + slot * 3;
    g[0] = g0; g[1] = g1; g[2] = g2;
    if (a.trace_g_theta) { float* o = a.trace_g_theta + (size_t)t * 66 + slot * 3; o[0] = g0; o[1] = g1; o[2] = g2; }
  } else {
    const int k = slot - NB;
    const float v = a.d_feat[(size_t)t * 200 + 189 + k];
    a.g_beta[(size_t)t * a.ld_gb + k] = v;
    if (a.trace_g_beta) a.trace_g_beta[(size_t)t * 10 + k] = v;
  }
}

hipError_t launch_rodrigues_bwd(const RodBwdArgs& a, hipStream_t stream) {
  const long n = (long)a.T * 32;
  hipLaunchKernelGGL(rodrigues_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// chain + skinning + sensors (+ reverse)
//
// Workgroup = CH_FRAMES frames.  Phase list (each a flat parallel-for over (frame, item), barrier in between):
//   P0  stage the packed index/weight tables (once per block) and each frame's rot | out row into LDS
//   P2  forward chain: (joint,row) walks the root path given as a bit mask (joints are topologically ordered)
//   P3  skinning of the needed vertices
//   P4a face normals, one (sensor,face) per lane      P4b per sensor: normal sum (in face order, as the reference),
//       frame, offsets, outputs, residual and its reverse down to (d centre, d helper, d face-normal)
//   P4c per (sensor,face): d e1, d e2                 P4d per vertex coordinate: gather its incident contributions
//   P5  d v_posed -> global;  per-bone force/moment partial sums over chunks of <= 16 (vertex,weight) pairs
//   P5c combine chunks per bone                       P6 subtree sums (bit mask)      P7 d R, d J -> global
// Every accumulation is a gather in a fixed order: results are bitwise reproducible and independent of the batch.
// ---------------------------------------------------------------------------------------------------------------
// Two shapes of the workgroup, picked by the launch (launch_chain_sensors): 4 frames on 256 threads from 4096 frames up
// (the 24 KB of tables are staged once per four frames and the narrow phases fill more lanes: 231 us against 253 for the
// other shape at T = 32768, 126 against 137 at 16384, 69 against 74 at 8192; whole forwards at 128 / 256 / 512 windows
// 1-2 % faster), 2 frames on 192 threads below (twice as many workgroups for the short launches of the streaming drivers).
// Measured and rejected: 3 x 192 (236), 3 x 256 (241), 4 x 192 (257), 4 x 224 (243), 5 x 256 (261), 8 x 512 (255),
// anything with more than 256 threads per workgroup (320-390 us: the register budget halves).

// (frame, item) of flat index i < FR * n without an integer division (n is a run-time value in most phases, and
// a division costs ~25 instructions in a kernel whose limit is instruction issue).
template <int FR>
__device__ __forceinline__ void frame_split_t(int i, int n, int& f, int& o) {
  static_assert(FR >= 1 && FR <= 8, "a chain of compares");
  f = 0;
#pragma unroll
  for (int k = 1; k < FR; ++k) f += i >= k * n ? 1 : 0;
  o = i - f * n;
}
constexpr int CHUNK = CHAIN_CHUNK;  // (vertex, weight) pairs per partial-sum chunk, lists padded with weight 0

size_t chain_lds_bytes(const SmplTables& tab, int frames_per_block) {
  const ChainLds l = chain_layout(tab.nv, tab.ncp, tab.max_deg, tab.n_chunks);
  return ((size_t)l.total * frames_per_block + ((tab.off.total + 3) & ~3)) * sizeof(float);
}

#ifdef EMPOSE_CHAIN_TRACE   // dev build only (scripts/dev/chain_trace.sh): shader-clock stamps of two blocks per phase
__device__ long long g_chain_trace[2][32];
#define CH_STAMP(i) \
  if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == 9000)) g_chain_trace[blockIdx.x != 0][(i)] = clock64();
#else
#define CH_STAMP(i)
#endif

template <int CH_FRAMES, int CH_THREADS>
__global__ __launch_bounds__(CH_THREADS) void chain_sensors_kernel(ChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  auto frame_split = [](int i, int n, int& f, int& o) { frame_split_t<CH_FRAMES>(i, n, f, o); };
  CH_STAMP(0)
  const SmplTables& tb = a.tab;
  const ChainTabs& O = tb.off;
  const ChainLds L = chain_layout(tb.nv, tb.ncp, tb.max_deg, tb.n_chunks);
  const int tid = threadIdx.x;
  const int t0 = blockIdx.x * CH_FRAMES;
  const int nf = min(CH_FRAMES, a.T - t0);
  const int nv3 = tb.nv * 3;
  const int md = tb.max_deg;
  const uint32_t md_magic = 65536u / (uint32_t)md + 1u;   // floor(x * magic / 65536) == x / md for x < 12 * md (md <= 64)
  const bool cot = a.cot_pos != nullptr;          // external cotangents (training) instead of the residual
  const bool bwd = a.tgt != nullptr || cot;
  float* frames = smem;
  const uint32_t* TI = reinterpret_cast<const uint32_t*>(smem + (size_t)L.total * CH_FRAMES);  // tables (ints)
  const float* TF = reinterpret_cast<const float*>(TI);                                        // tables (floats)

  // ---- P0: tables + per-frame rot | out.  Every global load of the phase (16-byte pieces of the table blob, the
  // frames' rot | out rows) is issued before the first LDS store, so the phase costs one round trip instead of one per
  // copy-loop iteration; tables or frame rows larger than one batch fall through to the batched loops below.
  {
    uint4* dst = reinterpret_cast<uint4*>(smem + (size_t)L.total * CH_FRAMES);
    const uint4* src = reinterpret_cast<const uint4*>(tb.blob);
    const int n16 = (O.total + 3) >> 2;   // the blob is padded to a multiple of 4 words
    const int row = NB * 9 + tb.ncp;
    const int n = nf * row;
    // 8 x 192 x 16 B = 24 KB of tables; CH_FRAMES frames x (198 + ncp ~ 320) floats over the workgroup's threads
    constexpr int TB = (8 * 192 + CH_THREADS - 1) / CH_THREADS, FB = (CH_FRAMES * 520 + CH_THREADS - 1) / CH_THREADS;
    auto frame_src = [&](int i) -> float {
      int f, o;
      frame_split(i, row, f, o);
      return o < NB * 9 ? a.rot[(size_t)(t0 + f) * (NB * 9) + o] : a.out[(size_t)(t0 + f) * tb.ncp + (o - NB * 9)];
    };
    auto frame_dst = [&](int i, float v) {
      int f, o;
      frame_split(i, row, f, o);
      frames[f * L.total + (o < NB * 9 ? L.rot + o : L.out + (o - NB * 9))] = v;
    };
    uint4 tv[TB];
    float fv[FB];
#pragma unroll
    for (int u = 0; u < TB; ++u) { const int i = tid + u * CH_THREADS; tv[u] = i < n16 ? src[i] : uint4{0u, 0u, 0u, 0u}; }
#pragma unroll
    for (int u = 0; u < FB; ++u) { const int i = tid + u * CH_THREADS; fv[u] = i < n ? frame_src(i) : 0.f; }
#pragma unroll
    for (int u = 0; u < TB; ++u) { const int i = tid + u * CH_THREADS; if (i < n16) dst[i] = tv[u]; }
#pragma unroll
    for (int u = 0; u < FB; ++u) { const int i = tid + u * CH_THREADS; if (i < n) frame_dst(i, fv[u]); }
    for (int i0 = tid + TB * CH_THREADS; i0 < n16; i0 += TB * CH_THREADS) {
#pragma unroll
      for (int u = 0; u < TB; ++u) { const int i = i0 + u * CH_THREADS; tv[u] = i < n16 ? src[i] : uint4{0u, 0u, 0u, 0u}; }
#pragma unroll
      for (int u = 0; u < TB; ++u) { const int i = i0 + u * CH_THREADS; if (i < n16) dst[i] = tv[u]; }
    }
    for (int i0 = tid + FB * CH_THREADS; i0 < n; i0 += FB * CH_THREADS) {
#pragma unroll
      for (int u = 0; u < FB; ++u) { const int i = i0 + u * CH_THREADS; fv[u] = i < n ? frame_src(i) : 0.f; }
#pragma unroll
      for (int u = 0; u < FB; ++u) { const int i = i0 + u * CH_THREADS; if (i < n) frame_dst(i, fv[u]); }
    }
  }
  __syncthreads();
  CH_STAMP(1)

  // ---- P2: forward chain
  for (int i = tid; i < nf * NB * 3; i += CH_THREADS) {
    int f, jr;
    frame_split(i, NB * 3, f, jr);
    const int j = jr / 3, r = jr % 3;
    float* S = frames + f * L.total;
    const float* sR = S + L.rot;
    const float* sJ = S + L.out + tb.j_off;
    float row0 = sR[r * 3 + 0], row1 = sR[r * 3 + 1], row2 = sR[r * 3 + 2];  // root rotation, row r
    float tr = sJ[r];
    int prev = 0;
    uint32_t pm = TI[O.path_mask + j] & ~1u;
    while (pm) {
      const int q = __ffs(pm) - 1;
      pm &= pm - 1;
      const float* Jq = sJ + q * 3;
      const float* Jp = sJ + prev * 3;
      tr = row0 * (Jq[0] - Jp[0]) + row1 * (Jq[1] - Jp[1]) + row2 * (Jq[2] - Jp[2]) + tr;
      const float* Rq = sR + q * 9;
      const float n0 = row0 * Rq[0] + row1 * Rq[3] + row2 * Rq[6];
      const float n1 = row0 * Rq[1] + row1 * Rq[4] + row2 * Rq[7];
      const float n2 = row0 * Rq[2] + row1 * Rq[5] + row2 * Rq[8];
      row0 = n0; row1 = n1; row2 = n2;
      prev = q;
    }
    float* G = S + L.g + j * 12;
    G[r * 3 + 0] = row0; G[r * 3 + 1] = row1; G[r * 3 + 2] = row2;
    G[9 + r] = tr;
    const float* Jj = sJ + j * 3;
    S[L.at + j * 3 + r] = tr - (row0 * Jj[0] + row1 * Jj[1] + row2 * Jj[2]);
    const size_t go = (size_t)(t0 + f) * 66 + j * 3 + r;
    a.joints[go] = tr;
    if (a.joints2) a.joints2[go] = tr;
  }
  __syncthreads();
  CH_STAMP(2)

  // ---- P3: linear blend skinning of the needed vertices, one (vertex, coordinate) per lane
  for (int i = tid; i < nf * nv3; i += CH_THREADS) {
    int f, sr;
    frame_split(i, nv3, f, sr);
    const int s = sr / 3, r = sr % 3;
    float* S = frames + f * L.total;
    const float* vp = S + L.out + s * 3;
    float T0 = 0.f, T1 = 0.f, T2 = 0.f, T3 = 0.f;  // row r of the blended 3x4 transform
#pragma unroll 4
    for (int k = 0; k < tb.kb; ++k) {  // padding entries have weight 0 / bone 0: no branch, loads can be batched
      const float w = TF[O.skin_w + s * tb.kb + k];
      const int b = TI[O.skin_idx + s * tb.kb + k];
      const float* G = S + L.g + b * 12 + r * 3;
      T0 += w * G[0]; T1 += w * G[1]; T2 += w * G[2];
      T3 += w * S[L.at + b * 3 + r];
    }
    S[L.v + sr] = T0 * vp[0] + T1 * vp[1] + T2 * vp[2] + T3;
  }
  __syncthreads();
  CH_STAMP(3)

  // ---- P4a: un-normalised face normals
  for (int i = tid; i < nf * 12 * md; i += CH_THREADS) {
    int f, mk;
    frame_split(i, 12 * md, f, mk);
    float* S = frames + f * L.total;
    const float* V = S + L.v;
    const uint32_t* fc = TI + O.s_faces + mk * 3;  // padding faces are (c,c,c): zero normal, zero cotangent
    const float* v0 = V + fc[0] * 3;
    const float* v1 = V + fc[1] * 3;
    const float* v2 = V + fc[2] * 3;
    const float e1[3] = {v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2]};
    const float e2[3] = {v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2]};
    cross3(e1, e2, S + L.fn + mk * 3);
  }
  __syncthreads();
  CH_STAMP(4)

  // ---- P4b: per sensor
  for (int i = tid; i < nf * 12; i += CH_THREADS) {
    int f, m;
    frame_split(i, 12, f, m);
    const int t = t0 + f;
    float* S = frames + f * L.total;
    const float* V = S + L.v;
    const int c = TI[O.s_center + m], h = TI[O.s_helper + m], deg = TI[O.s_deg + m];
    // Everything this sensor reads from global memory is requested up front: the stores of pos / ori further down
    // would otherwise sit between two dependent round trips (offsets, then targets).
    const int w = t / a.F;
    float Ro[9], to[3];
    {
      const float* pr = a.offset_r + ((size_t)w * 12 + m) * 9;
      const float* pt = a.offset_t + ((size_t)w * 12 + m) * 3;
#pragma unroll
      for (int k = 0; k < 9; ++k) Ro[k] = pr[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) to[k] = pt[k];
    }
    const int slot_m = a.used_slot[m];
    float tp[3] = {0.f, 0.f, 0.f}, tori[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, scale = 0.f;
    if (bwd && !cot && slot_m >= 0) {
      const float* p3 = a.tgt + (size_t)t * a.ld_tgt + slot_m * 3;
      const float* p9 = a.tgt + (size_t)t * a.ld_tgt + a.n_markers * 3 + slot_m * 9;
#pragma unroll
      for (int k = 0; k < 3; ++k) tp[k] = p3[k];
#pragma unroll
      for (int k = 0; k < 9; ++k) tori[k] = p9[k];
      scale = a.frame_scale[t];
    }
    float n[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < deg; ++k) {
      const float* fn = S + L.fn + (m * md + k) * 3;
      n[0] += fn[0]; n[1] += fn[1]; n[2] += fn[2];
    }
    // One (IEEE) reciprocal per normalisation, then multiplications: the per-sensor arithmetic runs on 24 of the 192
    // lanes, so its instruction count is the phase's latency.
    const float inv_deg = 1.f / (float)deg;
    n[0] *= inv_deg; n[1] *= inv_deg; n[2] *= inv_deg;
    const float inv_nn = 1.f / norm3(n);
    const float nh[3] = {n[0] * inv_nn, n[1] * inv_nn, n[2] * inv_nn};
    const float* vc = V + c * 3;
    const float* vh = V + h * 3;
    const float e[3] = {vh[0] - vc[0], vh[1] - vc[1], vh[2] - vc[2]};
    const float inv_ne = 1.f / norm3(e);
    const float sv[3] = {e[0] * inv_ne, e[1] * inv_ne, e[2] * inv_ne};
    float bb[3];
    cross3(nh, sv, bb);
    const float inv_nb = 1.f / norm3(bb);
    const float tv[3] = {bb[0] * inv_nb, bb[1] * inv_nb, bb[2] * inv_nb};
    float aa[3];
    cross3(tv, nh, aa);
    const float inv_na = 1.f / norm3(aa);
    const float s2[3] = {aa[0] * inv_na, aa[1] * inv_na, aa[2] * inv_na};
    // R_m columns (s2, tv, nh)
    const float Rm[9] = {s2[0], tv[0], nh[0], s2[1], tv[1], nh[1], s2[2], tv[2], nh[2]};
    float ori[9], pos[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int cc = 0; cc < 3; ++cc)
        ori[r * 3 + cc] = Rm[r * 3 + 0] * Ro[cc] + Rm[r * 3 + 1] * Ro[3 + cc] + Rm[r * 3 + 2] * Ro[6 + cc];
      pos[r] = vc[r] + (Rm[r * 3 + 0] * to[0] + Rm[r * 3 + 1] * to[1] + Rm[r * 3 + 2] * to[2]);
    }
    {
      float* po = a.pos + ((size_t)t * 12 + m) * 3;
      float* oo = a.ori + ((size_t)t * 12 + m) * 9;
      po[0] = pos[0]; po[1] = pos[1]; po[2] = pos[2];
#pragma unroll
      for (int k = 0; k < 9; ++k) oo[k] = ori[k];
      if (a.pos2) {
        float* p2 = a.pos2 + ((size_t)t * 12 + m) * 3;
        float* o2 = a.ori2 + ((size_t)t * 12 + m) * 9;
        p2[0] = pos[0]; p2[1] = pos[1]; p2[2] = pos[2];
#pragma unroll
        for (int k = 0; k < 9; ++k) o2[k] = ori[k];
      }
    }
    if (!bwd) continue;
    float* scr = S + L.scr + m * 9;  // d face-normal (3) | d centre (3) | d helper (3)
    float dpos[3], dori[9];
    if (cot) {
      const float* cp = a.cot_pos + ((size_t)t * 12 + m) * 3;
      const float* co = a.cot_ori + ((size_t)t * 12 + m) * 9;
      dpos[0] = cp[0]; dpos[1] = cp[1]; dpos[2] = cp[2];
#pragma unroll
      for (int k = 0; k < 9; ++k) dori[k] = co[k];
    } else {
      if (slot_m < 0 || scale == 0.f) {
#pragma unroll
        for (int k = 0; k < 9; ++k) scr[k] = 0.f;
        continue;
      }
      const float r0 = pos[0] - tp[0], r1 = pos[1] - tp[1], r2 = pos[2] - tp[2];
      const float sp = scale / sqrtf(r0 * r0 + r1 * r1 + r2 * r2);
      dpos[0] = r0 * sp; dpos[1] = r1 * sp; dpos[2] = r2 * sp;
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < 9; ++k) { dori[k] = ori[k] - tori[k]; q += dori[k] * dori[k]; }
      const float so = scale / sqrtf(q);
#pragma unroll
      for (int k = 0; k < 9; ++k) dori[k] *= so;
    }
    // dR_m = dori Ro^T + dpos (x) to
    float dRm[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int cc = 0; cc < 3; ++cc)
        dRm[r * 3 + cc] = dori[r * 3 + 0] * Ro[cc * 3 + 0] + dori[r * 3 + 1] * Ro[cc * 3 + 1] +
                          dori[r * 3 + 2] * Ro[cc * 3 + 2] + dpos[r] * to[cc];
    float ds2[3] = {dRm[0], dRm[3], dRm[6]};
    float dt[3] = {dRm[1], dRm[4], dRm[7]};
    float dnh[3] = {dRm[2], dRm[5], dRm[8]};
    float da[3], tmp[3];
    unit_bwd(ds2, s2, inv_na, da);   // a = t x nh
    cross3(nh, da, tmp); dt[0] += tmp[0]; dt[1] += tmp[1]; dt[2] += tmp[2];
    cross3(da, tv, tmp); dnh[0] += tmp[0]; dnh[1] += tmp[1]; dnh[2] += tmp[2];
    float db[3];
    unit_bwd(dt, tv, inv_nb, db);    // b = nh x s
    cross3(sv, db, tmp); dnh[0] += tmp[0]; dnh[1] += tmp[1]; dnh[2] += tmp[2];
    float dsv[3];
    cross3(db, nh, dsv);
    float de[3];
    unit_bwd(dsv, sv, inv_ne, de);
    float dn[3];
    unit_bwd(dnh, nh, inv_nn, dn);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      scr[k] = dn[k] * inv_deg;
      scr[3 + k] = dpos[k] - de[k];
      scr[6 + k] = de[k];
    }
  }
  if (!bwd) return;
  __syncthreads();
  CH_STAMP(5)

  // ---- P4c: per (sensor, face): cotangents of the two edge vectors
  for (int i = tid; i < nf * 12 * md; i += CH_THREADS) {
    int f, mk;
    frame_split(i, 12 * md, f, mk);
    const int m = (int)(((uint32_t)mk * md_magic) >> 16);   // mk / md for mk < 12 * md <= 4096
    float* S = frames + f * L.total;
    const float* V = S + L.v;
    const uint32_t* fc = TI + O.s_faces + mk * 3;
    const float* v0 = V + fc[0] * 3;
    const float* v1 = V + fc[1] * 3;
    const float* v2 = V + fc[2] * 3;
    const float e1[3] = {v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2]};
    const float e2[3] = {v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2]};
    const float* dfn = S + L.scr + m * 9;
    float* fg = S + L.fg + mk * 6;
    cross3(e2, dfn, fg);      // d e1
    cross3(dfn, e1, fg + 3);  // d e2
  }
  __syncthreads();
  CH_STAMP(6)

  // ---- P4d: gather the vertex cotangents (fixed order => deterministic).  Every incidence is one packed word
  // (a:13 | b:13 | use_b:1 | negate:1 | null:1, offsets into the frame's LDS record): contribution = +-(S[a+r] + S[b+r]);
  // lists are padded to a multiple of 4 with null words, so four independent LDS round trips are in flight.
  for (int i = tid; i < nf * nv3; i += CH_THREADS) {
    int f, sr;
    frame_split(i, nv3, f, sr);
    const int s = sr / 3, r = sr % 3;
    float* S = frames + f * L.total;
    float acc = 0.f;
    const int q0 = TI[O.inc_ptr + s], q1 = TI[O.inc_ptr + s + 1];
    for (int q = q0; q < q1; q += 4) {
      uint32_t code[4];
      float x[4], y[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) code[u] = TI[O.inc_code + q + u];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        x[u] = S[(code[u] & 0x1fffu) + r];
        y[u] = S[((code[u] >> 13) & 0x1fffu) + r];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float v = x[u] + ((code[u] >> 26) & 1u ? y[u] : 0.f);
        const float sg = (code[u] >> 28) & 1u ? 0.f : ((code[u] >> 27) & 1u ? -1.f : 1.f);
        acc += sg * v;
      }
    }
    S[L.dv + sr] = acc;
  }
  __syncthreads();
  CH_STAMP(7)

  // ---- P5: d v_posed (to global) and per-chunk force / world-space moment partial sums
  for (int i = tid; i < nf * tb.ncp; i += CH_THREADS) {
    int f, col;
    frame_split(i, tb.ncp, f, col);
    if (col >= tb.j_off && col < tb.j_off + NB * 3) continue;  // d J is written in P7
    float acc = 0.f;
    if (col < nv3) {
      const int s = col / 3, cc = col % 3;
      const float* S = frames + f * L.total;
      const float* dv = S + L.dv + s * 3;
#pragma unroll 4
      for (int k = 0; k < tb.kb; ++k) {
        const float w = TF[O.skin_w + s * tb.kb + k];
        const float* G = S + L.g + TI[O.skin_idx + s * tb.kb + k] * 12;
        acc += w * (G[0 + cc] * dv[0] + G[3 + cc] * dv[1] + G[6 + cc] * dv[2]);
      }
    }
    a.d_out[(size_t)(t0 + f) * tb.ncp + col] = acc;
  }
  for (int i = tid; i < nf * tb.n_chunks; i += CH_THREADS) {
    int f, ch;
    frame_split(i, tb.n_chunks, f, ch);
    float* S = frames + f * L.total;
    const int b = TI[O.chunk_bone + ch];
    const int q0 = TI[O.chunk_beg + ch];
    float G[12];
#pragma unroll
    for (int k = 0; k < 9; ++k) G[k] = S[L.g + b * 12 + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) G[9 + k] = S[L.at + b * 3 + k];
    float acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = 0.f;
#pragma unroll
    for (int q = 0; q < CHUNK; ++q) {  // chunks are padded to CHUNK pairs with weight 0
      const int sidx = TI[O.bone_vert + q0 + q];
      const float wq = TF[O.bone_w + q0 + q];
      const float* vp = S + L.out + sidx * 3;
      const float* dv = S + L.dv + sidx * 3;
      const float d0 = wq * dv[0], d1 = wq * dv[1], d2 = wq * dv[2];
      const float x0 = G[0] * vp[0] + G[1] * vp[1] + G[2] * vp[2] + G[9];
      const float x1 = G[3] * vp[0] + G[4] * vp[1] + G[5] * vp[2] + G[10];
      const float x2 = G[6] * vp[0] + G[7] * vp[1] + G[8] * vp[2] + G[11];
      acc[0] += d0 * x0; acc[1] += d0 * x1; acc[2] += d0 * x2;
      acc[3] += d1 * x0; acc[4] += d1 * x1; acc[5] += d1 * x2;
      acc[6] += d2 * x0; acc[7] += d2 * x1; acc[8] += d2 * x2;
      acc[9] += d0; acc[10] += d1; acc[11] += d2;
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) S[L.part + ch * 12 + k] = acc[k];
  }
  __syncthreads();
  CH_STAMP(8)

  // ---- P5c: per bone: sum its chunks (chunks of a bone are contiguous)
  for (int i = tid; i < nf * NB * 12; i += CH_THREADS) {
    int f, be;
    frame_split(i, NB * 12, f, be);
    const int b = be / 12, e = be % 12;
    float* S = frames + f * L.total;
    float acc = 0.f;
    for (int ch = TI[O.bone_chunk_ptr + b]; ch < (int)TI[O.bone_chunk_ptr + b + 1]; ++ch) acc += S[L.part + ch * 12 + e];
    if (a.cot_joints) {
      // joint j (position G_j^t) moves rigidly with its parent's frame: force d_j at point t_j on bone parent(j)
      const float* dj = a.cot_joints + (size_t)(t0 + f) * 66;
      for (int j = 1; j < NB; ++j) {
        if ((int)TI[O.parents + j] != b) continue;
        if (e < 9) acc += dj[j * 3 + e / 3] * S[L.g + j * 12 + 9 + e % 3];
        else acc += dj[j * 3 + (e - 9)];
      }
    }
    S[L.m + (int)TI[O.dfs_pos + b] * 12 + e] = acc;   // stored in depth-first order for the prefix sums below
  }
  __syncthreads();
  CH_STAMP(9)

  // ---- P6: subtree sums  X_j = sum_sub M_b - Fs_j (x) t_j.  With the bones in depth-first pre-order a subtree is a
  // contiguous range, so the sum is a difference of two prefix sums; the prefix runs in fp64 (22 additions per lane), so
  // the difference, rounded once to fp32, is the correctly rounded subtree sum.
  for (int i = tid; i < nf * 12; i += CH_THREADS) {
    int f, e;
    frame_split(i, 12, f, e);
    float* S = frames + f * L.total;
    double* P = reinterpret_cast<double*>(S + L.pd);
    float mv[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) mv[k] = S[L.m + k * 12 + e];
    double run = 0.0;
    P[e] = 0.0;
#pragma unroll
    for (int k = 0; k < NB; ++k) { run += (double)mv[k]; P[(k + 1) * 12 + e] = run; }
  }
  __syncthreads();
  for (int i = tid; i < nf * NB * 12; i += CH_THREADS) {
    int f, je;
    frame_split(i, NB * 12, f, je);
    const int j = je / 12, e = je % 12;
    float* S = frames + f * L.total;
    const double* P = reinterpret_cast<const double*>(S + L.pd);
    const int p0 = TI[O.dfs_pos + j], p1 = p0 + (int)TI[O.sub_size + j];
    const int ar = e < 9 ? e / 3 : e - 9, cc = e % 3;
    const float fs = (float)(P[p1 * 12 + 9 + ar] - P[p0 * 12 + 9 + ar]);
    if (e < 9) {
      const float ms = (float)(P[p1 * 12 + e] - P[p0 * 12 + e]);
      S[L.x + je] = ms - fs * S[L.g + j * 12 + 9 + cc];
    } else {
      S[L.x + je] = fs;
    }
  }
  __syncthreads();
  CH_STAMP(10)

  // ---- P7: d R_j = G_p^T X_j G_j  and  d J_j = (G_p - G_j)^T Fs_j
  for (int i = tid; i < nf * NB * 12; i += CH_THREADS) {
    int f, je;
    frame_split(i, NB * 12, f, je);
    const int j = je / 12, e = je % 12;
    const float* S = frames + f * L.total;
    const float* Gj = S + L.g + j * 12;
    const float* X = S + L.x + j * 12;
    const int p = (int)TI[O.parents + j];
    const float* Gp = S + L.g + (p < 0 ? 0 : p) * 12;
    if (e < 9) {
      const int r = e / 3, cc = e % 3;
      float acc = 0.f;
#pragma unroll
      for (int ar = 0; ar < 3; ++ar) {
        const float xg = X[ar * 3 + 0] * Gj[0 + cc] + X[ar * 3 + 1] * Gj[3 + cc] + X[ar * 3 + 2] * Gj[6 + cc];
        const float gp = p < 0 ? (ar == r ? 1.f : 0.f) : Gp[ar * 3 + r];
        acc += gp * xg;
      }
      a.d_rot[((size_t)(t0 + f) * NB + j) * 9 + e] = acc;
    } else {
      const int cc = e - 9;
      float acc = 0.f;
#pragma unroll
      for (int ar = 0; ar < 3; ++ar) {
        const float gp = p < 0 ? (ar == cc ? 1.f : 0.f) : Gp[ar * 3 + cc];
        acc += (gp - Gj[ar * 3 + cc]) * X[9 + ar];
        if (a.cot_joints) acc += gp * a.cot_joints[(size_t)(t0 + f) * 66 + j * 3 + ar];
      }
      a.d_out[(size_t)(t0 + f) * tb.ncp + tb.j_off + j * 3 + cc] = acc;
    }
  }
  CH_STAMP(11)
}

template <int FR, int NT>
static hipError_t launch_chain_cfg(const ChainArgs& a, hipStream_t stream) {
  const size_t lds = chain_lds_bytes(a.tab, FR);
  if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(chain_sensors_kernel<FR, NT>), lds)) return e;
  const int blocks = (a.T + FR - 1) / FR;
  hipLaunchKernelGGL((chain_sensors_kernel<FR, NT>), dim3(blocks), dim3(NT), lds, stream, a);
  return hipGetLastError();
}

hipError_t launch_chain_sensors(const ChainArgs& a, hipStream_t stream) {
  if (a.T >= 4096 && chain_lds_bytes(a.tab, 4) <= 64 * 1024) return launch_chain_cfg<4, 256>(a, stream);
  return launch_chain_cfg<2, 192>(a, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// Full mesh (final vertices): chain to relative transforms, then dense skinning of all V vertices.
// ---------------------------------------------------------------------------------------------------------------
__global__ void mesh_chain_kernel(MeshChainArgs a) {
  // one thread per (frame, joint): walks from the root down to its joint with the running transform (3 x 3 | t) in
  // registers.  (Round 6: a thread per (frame, joint, ROW) before -- three times the threads, each repeating the walk and
  // the ancestor look-ups for one row: 119 us per 16384 frames, a tenth of the full-mesh evaluation.)  Joints >= NB (the 30
  // hand joints of SMPL-H) have zero pose on this path (reference smpl.py:99), i.e. identity rotation under either
  // Rodrigues convention: they only translate along their parent's frame and get no skinning transform of their own
  // (weights folded into the wrists).
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int nj = a.n_joints;
  // (the parent table in LDS: the walks below are chains of dependent look-ups, ~80 per thread)
  __shared__ int parents_s[MESH_MAX_JOINTS];
  for (int i = threadIdx.x; i < nj; i += blockDim.x) parents_s[i] = a.parents[i];
  __syncthreads();
  if (idx >= a.T * nj) return;
  const int t = idx / nj, j = idx % nj;
  const float* R = a.rot + (size_t)t * NB * 9;
  const float* J = a.out + (size_t)t * a.ncp + a.j_off;
  // the joints from the root down to j, without a per-thread array of the path (a dynamically indexed array lives in
  // scratch memory): the ancestor k levels above j is found by walking up k times -- chains are at most a dozen joints long
  int n = 0;
  for (int q = j; q >= 0; q = parents_s[q]) ++n;
  float G[3][3], tr[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    G[r][0] = R[r * 3 + 0]; G[r][1] = R[r * 3 + 1]; G[r][2] = R[r * 3 + 2];
    tr[r] = J[r];
  }
  int prev = 0;
  for (int k = n - 2; k >= 0; --k) {
    int q = j;
    for (int up = 0; up < k; ++up) q = parents_s[q];
    const float* Jq = J + q * 3;
    const float* Jp = J + prev * 3;
    const float dx = Jq[0] - Jp[0], dy = Jq[1] - Jp[1], dz = Jq[2] - Jp[2];
#pragma unroll
    for (int r = 0; r < 3; ++r) tr[r] = G[r][0] * dx + G[r][1] * dy + G[r][2] * dz + tr[r];
    if (q < NB) {
      const float* Rq = R + q * 9;
      float Rv[9];
#pragma unroll
      for (int e = 0; e < 9; ++e) Rv[e] = Rq[e];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const float n0 = G[r][0] * Rv[0] + G[r][1] * Rv[3] + G[r][2] * Rv[6];
        const float n1 = G[r][0] * Rv[1] + G[r][1] * Rv[4] + G[r][2] * Rv[7];
        const float n2 = G[r][0] * Rv[2] + G[r][1] * Rv[5] + G[r][2] * Rv[8];
        G[r][0] = n0; G[r][1] = n1; G[r][2] = n2;
      }
    }
    prev = q;
  }
  if (j < NB) {
    const float* Jj = J + j * 3;
    // relative transform, row r: (G^R[r][0..2], A^t[r]) -- one 16-byte read per (bone, row) in the skinning epilogue
#pragma unroll
    for (int r = 0; r < 3; ++r)
      *reinterpret_cast<float4*>(a.xf + (((size_t)t * NB + j) * 3 + r) * 4) =
          make_float4(G[r][0], G[r][1], G[r][2], tr[r] - (G[r][0] * Jj[0] + G[r][1] * Jj[1] + G[r][2] * Jj[2]));
  }
#pragma unroll
  for (int r = 0; r < 3; ++r)
    a.joints[(size_t)t * nj * 3 + j * 3 + r] = tr[r] + (a.trans ? a.trans[(size_t)t * 3 + r] : 0.f);
}

hipError_t launch_mesh_chain(const MeshChainArgs& a, hipStream_t stream) {
  const long n = (long)a.T * a.n_joints;
  hipLaunchKernelGGL(mesh_chain_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a);
  return hipGetLastError();
}

// The dense skinning of all V vertices is mesh.hip's mesh_rows_kernel.

// ---------------------------------------------------------------------------------------------------------------
// Virtual sensors from arbitrary full-mesh vertices (reference virtual_sensors.py:16-38,77-96; utils.py:126-146):
// one lane per (frame, sensor).
// ---------------------------------------------------------------------------------------------------------------
__global__ void virtual_sensors_kernel(VirtualSensorArgs a) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= a.T * a.M) return;
  const int t = idx / a.M, m = idx % a.M;
  const float* V = a.vertices + (size_t)t * a.V * 3;
  const int deg = a.deg[m];
  const int* faces = a.faces + (size_t)m * a.max_deg * 3;
  float n[3] = {0.f, 0.f, 0.f};
  for (int k = 0; k < deg; ++k) {
    const float* v0 = V + (size_t)faces[k * 3 + 0] * 3;
    const float* v1 = V + (size_t)faces[k * 3 + 1] * 3;
    const float* v2 = V + (size_t)faces[k * 3 + 2] * 3;
    const float e1[3] = {v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2]};
    const float e2[3] = {v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2]};
    float fn[3];
    cross3(e1, e2, fn);
    n[0] += fn[0]; n[1] += fn[1]; n[2] += fn[2];
  }
  const float fdeg = (float)deg;
  n[0] /= fdeg; n[1] /= fdeg; n[2] /= fdeg;
  const float nn = norm3(n);
  const float nh[3] = {n[0] / nn, n[1] / nn, n[2] / nn};
  const float* vc = V + (size_t)a.center[m] * 3;
  const float* vh = V + (size_t)a.helper[m] * 3;
  const float e[3] = {vh[0] - vc[0], vh[1] - vc[1], vh[2] - vc[2]};
  const float ne = norm3(e);
  const float sv[3] = {e[0] / ne, e[1] / ne, e[2] / ne};
  float bb[3];
  cross3(nh, sv, bb);
  const float nb = norm3(bb);
  const float tv[3] = {bb[0] / nb, bb[1] / nb, bb[2] / nb};
  float aa[3];
  cross3(tv, nh, aa);
  const float na = norm3(aa);
  float* po = a.pos + (size_t)idx * 3;
  float* oo = a.ori + (size_t)idx * 9;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    po[r] = vc[r];
    oo[r * 3 + 0] = aa[r] / na;
    oo[r * 3 + 1] = tv[r];
    oo[r * 3 + 2] = nh[r];
    if (a.normals) a.normals[(size_t)idx * 3 + r] = n[r];
  }
}

hipError_t launch_virtual_sensors(const VirtualSensorArgs& a, hipStream_t stream) {
  const long n = (long)a.T * a.M;
  hipLaunchKernelGGL(virtual_sensors_kernel, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, stream, a);
  return hipGetLastError();
}

}  // namespace empose

#ifdef EMPOSE_CHAIN_TRACE
extern "C" int empose_debug_chain_trace(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(empose::g_chain_trace), sizeof(long long) * 64);
}
#endif
