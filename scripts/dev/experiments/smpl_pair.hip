// chain_pair_kernel: the per-frame part of one SMPL-H evaluation on the sensor sub-mesh -- kinematic chain, skinning,
// vertex normals and sensor frames, sensor offsets, reconstruction residual and the hand-derived reverse pass (the same
// mathematics as chain_sensors_kernel in smpl.hip; reference models.py:471-483,560-579, virtual_sensors.py:16-38,
// utils.py:126-146, loss.py:23-41; blueprint oracle/analytic_np.py) -- restructured around what the counters said
// about that kernel (profiles/r02_chain_counters.txt): it was bound by VALU ISSUE, 3290 wave-instructions per frame
// at 72 % of the issue slots, most of them index arithmetic, table decoding and 4-byte LDS traffic around ~15 k useful
// multiply-adds.  Here
//   * a lane works on TWO frames at once: every per-frame quantity is a float2 (frame a, frame b) and every
//     multiply-add a packed v_pk_fma_f32, so index arithmetic, table reads and address computation are paid once per
//     frame pair, and LDS traffic moves in 8- and 16-byte pieces;
//   * work items are fat: a lane owns a whole vertex (3 coordinates), a whole joint (3 x 3 rotation + translation) or a
//     whole bone (moment 3 x 3 + force) instead of one scalar of it, so weights and indices are fetched once per item
//     and intermediate products stay in registers;
//   * the per-bone moment is accumulated as N_b = sum w dv (x) v_posed and rotated once per bone
//     (M_b = N_b G_b^T + F_b (x) A_b^t) instead of transforming every (vertex, bone) pair; subtree sums are a 21-step
//     in-place sweep up the tree (children before parents) instead of fp64 prefix sums;
//   * workgroups are persistent: the index / weight tables are staged into LDS once, then the workgroup loops over
//     tiles of NPAIR frame pairs.
// Every accumulation runs in a fixed order over fixed operands: results are bitwise reproducible and do not depend on
// the batch, the tile or the frame a frame is paired with.
#include "kernels.h"

#include <algorithm>

namespace empose {

namespace cp {

typedef float v2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

#ifndef EMPOSE_PAIR_NT
#define EMPOSE_PAIR_NT 256
#endif
#ifndef EMPOSE_PAIR_NPAIR
#define EMPOSE_PAIR_NPAIR 2
#endif
#ifndef EMPOSE_PAIR_WAVES
#define EMPOSE_PAIR_WAVES 2
#endif
constexpr int NT = EMPOSE_PAIR_NT;         // threads per workgroup
constexpr int NPAIR = EMPOSE_PAIR_NPAIR;   // frame pairs per tile

__device__ __forceinline__ v2 fma2(v2 a, v2 b, v2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2 splat(float x) { return v2{x, x}; }
__device__ __forceinline__ v2 sqrt2(v2 a) { return v2{sqrtf(a.x), sqrtf(a.y)}; }
__device__ __forceinline__ v2 rcp2(v2 a) { return v2{1.f / a.x, 1.f / a.y}; }
__device__ __forceinline__ void cross3(const v2* a, const v2* b, v2* o) {
  o[0] = fma2(a[1], b[2], -(a[2] * b[1]));
  o[1] = fma2(a[2], b[0], -(a[0] * b[2]));
  o[2] = fma2(a[0], b[1], -(a[1] * b[0]));
}
__device__ __forceinline__ v2 dot3(const v2* a, const v2* b) { return fma2(a[2], b[2], fma2(a[1], b[1], a[0] * b[0])); }
// 1 / |a|: one v_rsq_f32 per frame (1 ulp) instead of a square root and an IEEE division
__device__ __forceinline__ v2 rsqrt2(v2 a) { return v2{__builtin_amdgcn_rsqf(a.x), __builtin_amdgcn_rsqf(a.y)}; }
__device__ __forceinline__ v2 inv_norm3(const v2* a) { return rsqrt2(dot3(a, a)); }
// y = x / |x|  ->  dx = (dy - y (y . dy)) / |x|
__device__ __forceinline__ void unit_bwd(const v2* dy, const v2* y, v2 inv_n, v2* dx) {
  const v2 d = dot3(dy, y);
  dx[0] = (dy[0] - y[0] * d) * inv_n;
  dx[1] = (dy[1] - y[1] * d) * inv_n;
  dx[2] = (dy[2] - y[2] * d) * inv_n;
}

}  // namespace cp

using namespace cp;

#ifdef EMPOSE_PAIR_TRACE   // dev build only (scripts/dev/pair_sweep.sh): shader-clock stamps of two blocks per phase
__device__ long long g_pair_trace[2][32];
#define CP_STAMP(i) \
  if (threadIdx.x == 0 && tile == (int)blockIdx.x && (blockIdx.x == 0 || blockIdx.x == 300)) g_pair_trace[blockIdx.x != 0][(i)] = clock64();
#else
#define CP_STAMP(i)
#endif

__global__ __launch_bounds__(cp::NT) __attribute__((amdgpu_waves_per_eu(EMPOSE_PAIR_WAVES, EMPOSE_PAIR_WAVES)))
void chain_pair_kernel(ChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  const SmplTables& tb = a.tab;
  const PairTabs& O = tb.poff;
  const PairLds L = pair_layout(tb.nv, tb.ncp, tb.max_deg, tb.n_chunks);
  const int tid = threadIdx.x;
  const int nv = tb.nv, md = tb.max_deg, kb = tb.kb, ncp = tb.ncp, j_off = tb.j_off;
  const bool cot = a.cot_pos != nullptr;          // external cotangents (training) instead of the residual
  const bool bwd = a.tgt != nullptr || cot;
  v2* const S0 = reinterpret_cast<v2*>(smem_f);
  const uint32_t* TI = reinterpret_cast<const uint32_t*>(S0 + (size_t)L.total * NPAIR);
  const float* TF = reinterpret_cast<const float*>(TI);

  // ---- tables: once per workgroup
  {
    uint4* dst = reinterpret_cast<uint4*>(S0 + (size_t)L.total * NPAIR);
    const uint4* src = reinterpret_cast<const uint4*>(tb.pair_blob);
    const int n16 = O.total >> 2;
    for (int i = tid; i < n16; i += NT) dst[i] = src[i];
  }

  const int tile_frames = 2 * NPAIR;
  const int ntiles = (a.T + tile_frames - 1) / tile_frames;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int t0 = tile * tile_frames;
    CP_STAMP(0)
    __syncthreads();   // the previous tile is done with its LDS record (and, first trip, nothing: tables land below)

    // ---- P0: this tile's rot | out rows, interleaved into (frame a, frame b) pairs: a wave copies whole frames
    // (coalesced reads, no index arithmetic beyond the lane offset).  Frames past T repeat frame T-1 (finite
    // arithmetic, stores are masked).
    {
      const int wave = tid >> 6, lane = tid & 63;
      for (int f = wave; f < tile_frames; f += NT / 64) {
        const int t = min(t0 + f, a.T - 1);
        float* P = reinterpret_cast<float*>(S0 + (size_t)(f >> 1) * L.total) + (f & 1);
        const float* src_r = a.rot + (size_t)t * (NB * 9);
        const float* src_o = a.out + (size_t)t * ncp;
        float r[4], o[5];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int k = lane + u * 64; r[u] = k < NB * 9 ? src_r[k] : 0.f; }
#pragma unroll
        for (int u = 0; u < 5; ++u) { const int k = lane + u * 64; o[u] = k < ncp ? src_o[k] : 0.f; }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int k = lane + u * 64; if (k < NB * 9) P[(L.rot + k) * 2] = r[u]; }
#pragma unroll
        for (int u = 0; u < 5; ++u) { const int k = lane + u * 64; if (k < ncp) P[(L.out + k) * 2] = o[u]; }
        for (int k = lane + 320; k < ncp; k += 64) P[(L.out + k) * 2] = src_o[k];   // larger sub-meshes
      }
    }
    __syncthreads();

    CP_STAMP(1)
    // ---- P2: forward chain, one (pair, joint) per lane: walks down the root path (joints are topologically ordered,
    // the path is a bit mask), G_j = G_parent [R_j | J_j - J_parent]
    if (tid >= 64) {
      // Meanwhile the other waves fetch what the per-sensor phase needs from global memory, per (frame, sensor): offset
      // rotation (9) | offset translation (3) | target position or its external cotangent (3) | target orientation or
      // cotangent (9) | frame weight -- into the area the edge cotangents take later.
      constexpr int SW = 25;
      const int w0 = t0 / a.F, r0 = t0 - w0 * a.F;
      for (int i = tid - 64; i < tile_frames * 12 * SW; i += NT - 64) {
        const int f = i / (12 * SW), r = i - f * (12 * SW), m = r / SW, k = r - m * SW;
        const int t = min(t0 + f, a.T - 1);
        int w = w0, rr = r0 + (t - t0);
        while (rr >= a.F) { rr -= a.F; ++w; }
        const int slot_m = a.used_slot[m];
        float val = 0.f;
        if (k < 9) val = a.offset_r[((size_t)w * 12 + m) * 9 + k];
        else if (k < 12) val = a.offset_t[((size_t)w * 12 + m) * 3 + (k - 9)];
        else if (cot) {
          if (k < 15) val = a.cot_pos[((size_t)t * 12 + m) * 3 + (k - 12)];
          else if (k < 24) val = a.cot_ori[((size_t)t * 12 + m) * 9 + (k - 15)];
        } else if (a.tgt != nullptr && slot_m >= 0) {
          if (k < 15) val = a.tgt[(size_t)t * a.ld_tgt + slot_m * 3 + (k - 12)];
          else if (k < 24) val = a.tgt[(size_t)t * a.ld_tgt + a.n_markers * 3 + slot_m * 9 + (k - 15)];
          else val = a.frame_scale[t];
        }
        float* P = reinterpret_cast<float*>(S0 + (size_t)(f >> 1) * L.total + L.fg);
        P[r * 2 + (f & 1)] = val;
      }
    }
    for (int i = tid; i < NPAIR * NB; i += NT) {
      const int p = i / NB, j = i - p * NB;
      v2* S = S0 + (size_t)p * L.total;
      const v2* sR = S + L.rot;
      const v2* sJ = S + L.out + j_off;
      v2 G[9], tr[3];
#pragma unroll
      for (int e = 0; e < 9; ++e) G[e] = sR[e];
#pragma unroll
      for (int r = 0; r < 3; ++r) tr[r] = sJ[r];
      int prev = 0;
      uint32_t pm = TI[O.path_mask + j] & ~1u;
      while (pm) {
        const int q = __ffs(pm) - 1;
        pm &= pm - 1;
        const v2 d0 = sJ[q * 3 + 0] - sJ[prev * 3 + 0];
        const v2 d1 = sJ[q * 3 + 1] - sJ[prev * 3 + 1];
        const v2 d2 = sJ[q * 3 + 2] - sJ[prev * 3 + 2];
        v2 Rq[9];
#pragma unroll
        for (int e = 0; e < 9; ++e) Rq[e] = sR[q * 9 + e];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const v2 g0 = G[r * 3 + 0], g1 = G[r * 3 + 1], g2 = G[r * 3 + 2];
          tr[r] = fma2(g0, d0, fma2(g1, d1, fma2(g2, d2, tr[r])));
          G[r * 3 + 0] = fma2(g0, Rq[0], fma2(g1, Rq[3], g2 * Rq[6]));
          G[r * 3 + 1] = fma2(g0, Rq[1], fma2(g1, Rq[4], g2 * Rq[7]));
          G[r * 3 + 2] = fma2(g0, Rq[2], fma2(g1, Rq[5], g2 * Rq[8]));
        }
        prev = q;
      }
      v2* X = S + L.xf + j * 16;
      const v2 J0 = sJ[j * 3 + 0], J1 = sJ[j * 3 + 1], J2 = sJ[j * 3 + 2];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        X[r * 4 + 0] = G[r * 3 + 0]; X[r * 4 + 1] = G[r * 3 + 1]; X[r * 4 + 2] = G[r * 3 + 2];
        X[r * 4 + 3] = tr[r] - fma2(G[r * 3 + 0], J0, fma2(G[r * 3 + 1], J1, G[r * 3 + 2] * J2));
        X[12 + r] = tr[r];
      }
      const int ta = t0 + 2 * p, tbf = ta + 1;
      if (ta < a.T) {
        float* o = a.joints + (size_t)ta * 66 + j * 3;
        o[0] = tr[0].x; o[1] = tr[1].x; o[2] = tr[2].x;
        if (a.joints2) { float* o2 = a.joints2 + (size_t)ta * 66 + j * 3; o2[0] = tr[0].x; o2[1] = tr[1].x; o2[2] = tr[2].x; }
      }
      if (tbf < a.T) {
        float* o = a.joints + (size_t)tbf * 66 + j * 3;
        o[0] = tr[0].y; o[1] = tr[1].y; o[2] = tr[2].y;
        if (a.joints2) { float* o2 = a.joints2 + (size_t)tbf * 66 + j * 3; o2[0] = tr[0].y; o2[1] = tr[1].y; o2[2] = tr[2].y; }
      }
    }
    __syncthreads();

    CP_STAMP(2)
    // ---- P3: linear blend skinning, one (pair, vertex) per lane: blended 3 x 4 transform, then one mat-vec (the order
    // of the reference's lbs)
    for (int i = tid; i < NPAIR * nv; i += NT) {
      const int p = i / nv, s = i - p * nv;
      v2* S = S0 + (size_t)p * L.total;
      v2 T[12];
#pragma unroll
      for (int e = 0; e < 12; ++e) T[e] = splat(0.f);
      if (kb == 4) {   // the usual case (at most four bones per vertex): all loads of the item in flight at once
        const uint4 ea = *reinterpret_cast<const uint4*>(TI + O.skin + s * 8);
        const uint4 eb = *reinterpret_cast<const uint4*>(TI + O.skin + s * 8 + 4);
        const uint32_t bi[4] = {ea.x, ea.z, eb.x, eb.z};
        const float wf[4] = {__uint_as_float(ea.y), __uint_as_float(ea.w), __uint_as_float(eb.y), __uint_as_float(eb.w)};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const v2 w = splat(wf[k]);
          const v2* X = S + L.xf + bi[k] * 16;
#pragma unroll
          for (int c = 0; c < 12; ++c) T[c] = fma2(w, X[c], T[c]);
        }
      } else {
        for (int k = 0; k < kb; ++k) {   // padding entries have weight 0 / bone 0
          const uint2 e = *reinterpret_cast<const uint2*>(TI + O.skin + (s * kb + k) * 2);
          const v2 w = splat(__uint_as_float(e.y));
          const v2* X = S + L.xf + e.x * 16;
#pragma unroll
          for (int c = 0; c < 12; ++c) T[c] = fma2(w, X[c], T[c]);
        }
      }
      const v2* vp = S + L.out + s * 3;
      const v2 x = vp[0], y = vp[1], z = vp[2];
      v2* vo = S + L.v + s * 3;
#pragma unroll
      for (int r = 0; r < 3; ++r) vo[r] = fma2(T[r * 4 + 0], x, fma2(T[r * 4 + 1], y, fma2(T[r * 4 + 2], z, T[r * 4 + 3])));
    }
    __syncthreads();

    CP_STAMP(3)
    // ---- P4a: un-normalised face normals, one (pair, sensor, face) per lane
    for (int i = tid; i < NPAIR * 12 * md; i += NT) {
      const int p = i / (12 * md), mk = i - p * (12 * md);
      v2* S = S0 + (size_t)p * L.total;
      const v2* V = S + L.v;
      const uint4 fc = *reinterpret_cast<const uint4*>(TI + O.faces + mk * 4);
      const v2* v0 = V + fc.x * 3;
      const v2* v1 = V + fc.y * 3;
      const v2* v2p = V + fc.z * 3;
      const v2 e1[3] = {v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2]};
      const v2 e2[3] = {v2p[0] - v0[0], v2p[1] - v0[1], v2p[2] - v0[2]};
      cross3(e1, e2, S + L.fn + mk * 3);
    }
    __syncthreads();

    CP_STAMP(4)
    // ---- P4b: per (pair, sensor): normal (sum in face order, as the reference), frame, offsets, outputs, residual and
    // its reverse down to (d face-normal, d centre, d helper)
    for (int i = tid; i < NPAIR * 12; i += NT) {
      const int p = i / 12, m = i - p * 12;
      v2* S = S0 + (size_t)p * L.total;
      const v2* V = S + L.v;
      const int ta = t0 + 2 * p;
      const int c = TI[O.s_center + m], h = TI[O.s_helper + m], deg = TI[O.s_deg + m];
      const v2* sin_ = S + L.fg + m * 25;   // staged in P0
      v2 Ro[9], to[3];
#pragma unroll
      for (int k = 0; k < 9; ++k) Ro[k] = sin_[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) to[k] = sin_[9 + k];
      const int slot_m = a.used_slot[m];
      v2 n[3] = {splat(0.f), splat(0.f), splat(0.f)};
      for (int k = 0; k < deg; ++k) {
        const v2* fn = S + L.fn + (m * md + k) * 3;
        n[0] += fn[0]; n[1] += fn[1]; n[2] += fn[2];
      }
      const v2 inv_deg = splat(1.f / (float)deg);
      n[0] *= inv_deg; n[1] *= inv_deg; n[2] *= inv_deg;
      const v2 inv_nn = inv_norm3(n);
      const v2 nh[3] = {n[0] * inv_nn, n[1] * inv_nn, n[2] * inv_nn};
      const v2* vc = V + c * 3;
      const v2* vh = V + h * 3;
      const v2 e[3] = {vh[0] - vc[0], vh[1] - vc[1], vh[2] - vc[2]};
      const v2 inv_ne = inv_norm3(e);
      const v2 sv[3] = {e[0] * inv_ne, e[1] * inv_ne, e[2] * inv_ne};
      v2 bb[3];
      cross3(nh, sv, bb);
      const v2 inv_nb = inv_norm3(bb);
      const v2 tv[3] = {bb[0] * inv_nb, bb[1] * inv_nb, bb[2] * inv_nb};
      v2 aa[3];
      cross3(tv, nh, aa);
      const v2 inv_na = inv_norm3(aa);
      const v2 s2[3] = {aa[0] * inv_na, aa[1] * inv_na, aa[2] * inv_na};
      // R_m columns (s2, tv, nh)
      const v2 Rm[9] = {s2[0], tv[0], nh[0], s2[1], tv[1], nh[1], s2[2], tv[2], nh[2]};
      v2 ori[9], pos[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int cc = 0; cc < 3; ++cc)
          ori[r * 3 + cc] = fma2(Rm[r * 3 + 0], Ro[cc], fma2(Rm[r * 3 + 1], Ro[3 + cc], Rm[r * 3 + 2] * Ro[6 + cc]));
        pos[r] = vc[r] + fma2(Rm[r * 3 + 0], to[0], fma2(Rm[r * 3 + 1], to[1], Rm[r * 3 + 2] * to[2]));
      }
      if (ta < a.T) {
        float* po = a.pos + ((size_t)ta * 12 + m) * 3;
        float* oo = a.ori + ((size_t)ta * 12 + m) * 9;
        po[0] = pos[0].x; po[1] = pos[1].x; po[2] = pos[2].x;
#pragma unroll
        for (int k = 0; k < 9; ++k) oo[k] = ori[k].x;
        if (a.pos2) {
          float* p2 = a.pos2 + ((size_t)ta * 12 + m) * 3;
          float* o2 = a.ori2 + ((size_t)ta * 12 + m) * 9;
          p2[0] = pos[0].x; p2[1] = pos[1].x; p2[2] = pos[2].x;
#pragma unroll
          for (int k = 0; k < 9; ++k) o2[k] = ori[k].x;
        }
      }
      if (ta + 1 < a.T) {
        float* po = a.pos + ((size_t)(ta + 1) * 12 + m) * 3;
        float* oo = a.ori + ((size_t)(ta + 1) * 12 + m) * 9;
        po[0] = pos[0].y; po[1] = pos[1].y; po[2] = pos[2].y;
#pragma unroll
        for (int k = 0; k < 9; ++k) oo[k] = ori[k].y;
        if (a.pos2) {
          float* p2 = a.pos2 + ((size_t)(ta + 1) * 12 + m) * 3;
          float* o2 = a.ori2 + ((size_t)(ta + 1) * 12 + m) * 9;
          p2[0] = pos[0].y; p2[1] = pos[1].y; p2[2] = pos[2].y;
#pragma unroll
          for (int k = 0; k < 9; ++k) o2[k] = ori[k].y;
        }
      }
      if (!bwd) continue;
      v2* scr = S + L.scr + m * 9;  // d face-normal (3) | d centre (3) | d helper (3)
      v2 dpos[3], dori[9];
      if (cot) {
#pragma unroll
        for (int k = 0; k < 3; ++k) dpos[k] = sin_[12 + k];
#pragma unroll
        for (int k = 0; k < 9; ++k) dori[k] = sin_[15 + k];
      } else {
        const v2 scale = sin_[24];
        // a frame with weight 0 (padding, missing sensor) or an unused sensor contributes exactly zero: its residual
        // direction r / |r| may be 0 / 0 (all-zero padded inputs), so it is selected away, not multiplied away
        const bool live_a = slot_m >= 0 && scale.x != 0.f, live_b = slot_m >= 0 && scale.y != 0.f;
        const v2 r0 = pos[0] - sin_[12], r1 = pos[1] - sin_[13], r2 = pos[2] - sin_[14];
        const v2 sp = scale * rsqrt2(fma2(r0, r0, fma2(r1, r1, r2 * r2)));
        dpos[0] = r0 * sp; dpos[1] = r1 * sp; dpos[2] = r2 * sp;
        v2 q = splat(0.f);
#pragma unroll
        for (int k = 0; k < 9; ++k) { dori[k] = ori[k] - sin_[15 + k]; q = fma2(dori[k], dori[k], q); }
        const v2 so = scale * rsqrt2(q);
#pragma unroll
        for (int k = 0; k < 9; ++k) dori[k] *= so;
        if (!live_a) {
#pragma unroll
          for (int k = 0; k < 3; ++k) dpos[k].x = 0.f;
#pragma unroll
          for (int k = 0; k < 9; ++k) dori[k].x = 0.f;
        }
        if (!live_b) {
#pragma unroll
          for (int k = 0; k < 3; ++k) dpos[k].y = 0.f;
#pragma unroll
          for (int k = 0; k < 9; ++k) dori[k].y = 0.f;
        }
      }
      // dR_m = dori Ro^T + dpos (x) to
      v2 dRm[9];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int cc = 0; cc < 3; ++cc)
          dRm[r * 3 + cc] = fma2(dori[r * 3 + 0], Ro[cc * 3 + 0], fma2(dori[r * 3 + 1], Ro[cc * 3 + 1],
                                 fma2(dori[r * 3 + 2], Ro[cc * 3 + 2], dpos[r] * to[cc])));
      v2 ds2[3] = {dRm[0], dRm[3], dRm[6]};
      v2 dt[3] = {dRm[1], dRm[4], dRm[7]};
      v2 dnh[3] = {dRm[2], dRm[5], dRm[8]};
      v2 da[3], tmp[3];
      unit_bwd(ds2, s2, inv_na, da);   // a = t x nh
      cross3(nh, da, tmp); dt[0] += tmp[0]; dt[1] += tmp[1]; dt[2] += tmp[2];
      cross3(da, tv, tmp); dnh[0] += tmp[0]; dnh[1] += tmp[1]; dnh[2] += tmp[2];
      v2 db[3];
      unit_bwd(dt, tv, inv_nb, db);    // b = nh x s
      cross3(sv, db, tmp); dnh[0] += tmp[0]; dnh[1] += tmp[1]; dnh[2] += tmp[2];
      v2 dsv[3];
      cross3(db, nh, dsv);
      v2 de[3];
      unit_bwd(dsv, sv, inv_ne, de);
      v2 dn[3];
      unit_bwd(dnh, nh, inv_nn, dn);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        scr[k] = dn[k] * inv_deg;
        scr[3 + k] = dpos[k] - de[k];
        scr[6 + k] = de[k];
      }
    }
    if (!bwd) continue;   // forward only: next tile (the barrier at the top of the loop orders the LDS reuse)
    __syncthreads();

    CP_STAMP(5)
    // ---- P4c: per (pair, sensor, face): cotangents of the two edge vectors
    for (int i = tid; i < NPAIR * 12 * md; i += NT) {
      const int p = i / (12 * md), mk = i - p * (12 * md);
      const int m = mk / md;
      v2* S = S0 + (size_t)p * L.total;
      const v2* V = S + L.v;
      const uint4 fc = *reinterpret_cast<const uint4*>(TI + O.faces + mk * 4);
      const v2* v0 = V + fc.x * 3;
      const v2* v1 = V + fc.y * 3;
      const v2* v2p = V + fc.z * 3;
      const v2 e1[3] = {v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2]};
      const v2 e2[3] = {v2p[0] - v0[0], v2p[1] - v0[1], v2p[2] - v0[2]};
      const v2* dfn = S + L.scr + m * 9;
      v2* fg = S + L.fg + mk * 6;
      cross3(e2, dfn, fg);      // d e1
      cross3(dfn, e1, fg + 3);  // d e2
    }
    __syncthreads();

    CP_STAMP(6)
    // ---- P4d: vertex cotangents, one (pair, vertex) per lane, gathered in a fixed order from packed incidence words
    // (a:13 | b:13 | use_b:1 | negate:1 | null:1, float2 offsets into the pair's record): +-(S[a + r] + S[b + r]).
    // The positions are dead by now: dv overwrites them in place -- but another lane's gather may still read a position?
    // No: P4d reads only scr / fg; positions were last read in P4c (barrier above).
    for (int i = tid; i < NPAIR * nv; i += NT) {
      const int p = i / nv, s = i - p * nv;
      v2* S = S0 + (size_t)p * L.total;
      v2 acc[3] = {splat(0.f), splat(0.f), splat(0.f)};
      const int q0 = TI[O.inc_ptr + s], q1 = TI[O.inc_ptr + s + 1];
      for (int q = q0; q < q1; q += 4) {
        const uint4 code4 = *reinterpret_cast<const uint4*>(TI + O.inc_code + q);
        const uint32_t code[4] = {code4.x, code4.y, code4.z, code4.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const v2* xa = S + (code[u] & 0x1fffu);
          const v2* xb = S + ((code[u] >> 13) & 0x1fffu);
          const float ub = (code[u] >> 26) & 1u ? 1.f : 0.f;
          const float sg = (code[u] >> 28) & 1u ? 0.f : ((code[u] >> 27) & 1u ? -1.f : 1.f);
          const v2 sgv = splat(sg), ubv = splat(sg * ub);
#pragma unroll
          for (int r = 0; r < 3; ++r) acc[r] = fma2(ubv, xb[r], fma2(sgv, xa[r], acc[r]));
        }
      }
      v2* dv = S + L.v + s * 3;
      dv[0] = acc[0]; dv[1] = acc[1]; dv[2] = acc[2];
    }
    __syncthreads();

    CP_STAMP(7)
    // ---- P5a: d v_posed -> global, one (pair, vertex) per lane;  P5b: per-chunk partial sums of
    // N_b = sum w dv (x) v_posed (3 x 3) and F_b = sum w dv over CHAIN_CHUNK (vertex, weight) pairs of a bone
    for (int i = tid; i < NPAIR * nv; i += NT) {
      const int p = i / nv, s = i - p * nv;
      const v2* S = S0 + (size_t)p * L.total;
      const v2* dv = S + L.v + s * 3;
      const v2 d0 = dv[0], d1 = dv[1], d2 = dv[2];
      v2 acc[3] = {splat(0.f), splat(0.f), splat(0.f)};
      if (kb == 4) {
        const uint4 ea = *reinterpret_cast<const uint4*>(TI + O.skin + s * 8);
        const uint4 eb = *reinterpret_cast<const uint4*>(TI + O.skin + s * 8 + 4);
        const uint32_t bi[4] = {ea.x, ea.z, eb.x, eb.z};
        const float wf[4] = {__uint_as_float(ea.y), __uint_as_float(ea.w), __uint_as_float(eb.y), __uint_as_float(eb.w)};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const v2 w = splat(wf[k]);
          const v2* X = S + L.xf + bi[k] * 16;
#pragma unroll
          for (int cc = 0; cc < 3; ++cc)
            acc[cc] = fma2(w, fma2(X[cc], d0, fma2(X[4 + cc], d1, X[8 + cc] * d2)), acc[cc]);
        }
      } else {
        for (int k = 0; k < kb; ++k) {
          const uint2 e = *reinterpret_cast<const uint2*>(TI + O.skin + (s * kb + k) * 2);
          const v2 w = splat(__uint_as_float(e.y));
          const v2* X = S + L.xf + e.x * 16;
#pragma unroll
          for (int cc = 0; cc < 3; ++cc)
            acc[cc] = fma2(w, fma2(X[cc], d0, fma2(X[4 + cc], d1, X[8 + cc] * d2)), acc[cc]);
        }
      }
      const int ta = t0 + 2 * p;
      if (ta < a.T) { float* o = a.d_out + (size_t)ta * ncp + s * 3; o[0] = acc[0].x; o[1] = acc[1].x; o[2] = acc[2].x; }
      if (ta + 1 < a.T) { float* o = a.d_out + (size_t)(ta + 1) * ncp + s * 3; o[0] = acc[0].y; o[1] = acc[1].y; o[2] = acc[2].y; }
    }
    // padding columns of d_out (between the vertex and the joint block, after the joint block): zero
    {
      const int npad = ncp - nv * 3 - NB * 3;
      for (int i = tid; i < tile_frames * npad; i += NT) {
        const int f = i / npad, k = i - f * npad;
        const int col = k < j_off - nv * 3 ? nv * 3 + k : j_off + NB * 3 + (k - (j_off - nv * 3));
        if (t0 + f < a.T) a.d_out[(size_t)(t0 + f) * ncp + col] = 0.f;
      }
    }
    for (int i = tid; i < NPAIR * tb.n_chunks; i += NT) {
      const int p = i / tb.n_chunks, ch = i - p * tb.n_chunks;
      v2* S = S0 + (size_t)p * L.total;
      v2 acc[12];
#pragma unroll
      for (int k = 0; k < 12; ++k) acc[k] = splat(0.f);
#pragma unroll
      for (int q = 0; q < CHAIN_CHUNK; ++q) {   // chunks are padded to CHAIN_CHUNK pairs with weight 0
        const uint2 e = *reinterpret_cast<const uint2*>(TI + O.chunk_ent + (ch * CHAIN_CHUNK + q) * 2);
        const v2 w = splat(__uint_as_float(e.y));
        const v2* vp = S + L.out + e.x * 3;
        const v2* dv = S + L.v + e.x * 3;
        const v2 d0 = w * dv[0], d1 = w * dv[1], d2 = w * dv[2];
        const v2 x0 = vp[0], x1 = vp[1], x2 = vp[2];
        acc[0] = fma2(d0, x0, acc[0]); acc[1] = fma2(d0, x1, acc[1]); acc[2] = fma2(d0, x2, acc[2]);
        acc[3] = fma2(d1, x0, acc[3]); acc[4] = fma2(d1, x1, acc[4]); acc[5] = fma2(d1, x2, acc[5]);
        acc[6] = fma2(d2, x0, acc[6]); acc[7] = fma2(d2, x1, acc[7]); acc[8] = fma2(d2, x2, acc[8]);
        acc[9] += d0; acc[10] += d1; acc[11] += d2;
      }
      v2* po = S + L.part + ch * 12;
#pragma unroll
      for (int k = 0; k < 12; ++k) po[k] = acc[k];
    }
    __syncthreads();

    CP_STAMP(8)
    // ---- P5c: per (pair, bone): sum its chunks (contiguous), rotate the moment once:
    // M_b = N_b G_b^RT + F_b (x) A_b^t  (= sum w dv (x) (G_b v_posed + A_b^t));  external joint cotangents act as a force
    // d_j at point t_j on the parent's bone
    for (int i = tid; i < NPAIR * NB; i += NT) {
      const int p = i / NB, b = i - p * NB;
      v2* S = S0 + (size_t)p * L.total;
      v2 Nf[12];
#pragma unroll
      for (int k = 0; k < 12; ++k) Nf[k] = splat(0.f);
      for (int ch = TI[O.bone_chunk_ptr + b]; ch < (int)TI[O.bone_chunk_ptr + b + 1]; ++ch) {
        const v2* pp = S + L.part + ch * 12;
#pragma unroll
        for (int k = 0; k < 12; ++k) Nf[k] += pp[k];
      }
      const v2* X = S + L.xf + b * 16;
      v2 M[12];
#pragma unroll
      for (int ar = 0; ar < 3; ++ar) {
#pragma unroll
        for (int cc = 0; cc < 3; ++cc)
          M[ar * 3 + cc] = fma2(Nf[ar * 3 + 0], X[cc * 4 + 0], fma2(Nf[ar * 3 + 1], X[cc * 4 + 1],
                                fma2(Nf[ar * 3 + 2], X[cc * 4 + 2], Nf[9 + ar] * X[cc * 4 + 3])));
        M[9 + ar] = Nf[9 + ar];
      }
      if (a.cot_joints) {
        const int ta = t0 + 2 * p;
        const float* dja = a.cot_joints + (size_t)min(ta, a.T - 1) * 66;
        const float* djb = a.cot_joints + (size_t)min(ta + 1, a.T - 1) * 66;
        for (int j = 1; j < NB; ++j) {
          if ((int)TI[O.parents + j] != b) continue;
          const v2* Xj = S + L.xf + j * 16;
#pragma unroll
          for (int ar = 0; ar < 3; ++ar) {
            const v2 d = v2{dja[j * 3 + ar], djb[j * 3 + ar]};
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) M[ar * 3 + cc] = fma2(d, Xj[12 + cc], M[ar * 3 + cc]);
            M[9 + ar] += d;
          }
        }
      }
      v2* mo = S + L.m + b * 12;
#pragma unroll
      for (int k = 0; k < 12; ++k) mo[k] = M[k];
    }
    __syncthreads();

    CP_STAMP(9)
    CP_STAMP(10)
    // ---- P6 + P7: per (pair, joint): subtree sums, X_j = Ms_j - Fs_j (x) t_j;  d R_j = G_p^T X_j G_j;  d J_j = (G_p - G_j)^T Fs_j
    for (int i = tid; i < NPAIR * NB; i += NT) {
      const int p = i / NB, j = i - p * NB;
      const v2* S = S0 + (size_t)p * L.total;
      const v2* Xj = S + L.xf + j * 16;
      const int pj = (int)TI[O.parents + j];
      // subtree sums (moment 9 | force 3): a gather over the subtree's bit mask in ascending bone order
      v2 Mj[12];
#pragma unroll
      for (int k = 0; k < 12; ++k) Mj[k] = splat(0.f);
      {
        uint32_t sm = TI[O.sub_mask + j];
        while (sm) {
          const int b0 = __ffs(sm) - 1;
          sm &= sm - 1;
          const v2* mb = S + L.m + b0 * 12;
          if (sm) {   // two members per trip: their loads overlap
            const int b1 = __ffs(sm) - 1;
            sm &= sm - 1;
            const v2* mc = S + L.m + b1 * 12;
#pragma unroll
            for (int k = 0; k < 12; ++k) Mj[k] += mb[k];
#pragma unroll
            for (int k = 0; k < 12; ++k) Mj[k] += mc[k];
          } else {
#pragma unroll
            for (int k = 0; k < 12; ++k) Mj[k] += mb[k];
          }
        }
      }
      const v2 Fs[3] = {Mj[9], Mj[10], Mj[11]};
      v2 XG[9];
#pragma unroll
      for (int ar = 0; ar < 3; ++ar) {
        const v2 x0 = fma2(-Fs[ar], Xj[12 + 0], Mj[ar * 3 + 0]);
        const v2 x1 = fma2(-Fs[ar], Xj[12 + 1], Mj[ar * 3 + 1]);
        const v2 x2 = fma2(-Fs[ar], Xj[12 + 2], Mj[ar * 3 + 2]);
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) XG[ar * 3 + cc] = fma2(x0, Xj[cc], fma2(x1, Xj[4 + cc], x2 * Xj[8 + cc]));
      }
      v2 dR[9], dJ[3];
      if (pj < 0) {
#pragma unroll
        for (int e = 0; e < 9; ++e) dR[e] = XG[e];
#pragma unroll
        for (int cc = 0; cc < 3; ++cc)
          dJ[cc] = Fs[cc] - fma2(Xj[cc], Fs[0], fma2(Xj[4 + cc], Fs[1], Xj[8 + cc] * Fs[2]));
      } else {
        const v2* Xp = S + L.xf + pj * 16;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int cc = 0; cc < 3; ++cc)
            dR[r * 3 + cc] = fma2(Xp[r], XG[cc], fma2(Xp[4 + r], XG[3 + cc], Xp[8 + r] * XG[6 + cc]));
#pragma unroll
        for (int cc = 0; cc < 3; ++cc)
          dJ[cc] = fma2(Xp[cc] - Xj[cc], Fs[0], fma2(Xp[4 + cc] - Xj[4 + cc], Fs[1], (Xp[8 + cc] - Xj[8 + cc]) * Fs[2]));
      }
      const int ta = t0 + 2 * p;
      if (a.cot_joints) {
        const float* dja = a.cot_joints + (size_t)min(ta, a.T - 1) * 66 + j * 3;
        const float* djb = a.cot_joints + (size_t)min(ta + 1, a.T - 1) * 66 + j * 3;
        const v2 d[3] = {v2{dja[0], djb[0]}, v2{dja[1], djb[1]}, v2{dja[2], djb[2]}};
        if (pj < 0) {
#pragma unroll
          for (int cc = 0; cc < 3; ++cc) dJ[cc] += d[cc];
        } else {
          const v2* Xp = S + L.xf + pj * 16;
#pragma unroll
          for (int cc = 0; cc < 3; ++cc) dJ[cc] = fma2(Xp[cc], d[0], fma2(Xp[4 + cc], d[1], fma2(Xp[8 + cc], d[2], dJ[cc])));
        }
      }
      if (ta < a.T) {
        float* o = a.d_rot + ((size_t)ta * NB + j) * 9;
#pragma unroll
        for (int e = 0; e < 9; ++e) o[e] = dR[e].x;
        float* oj = a.d_out + (size_t)ta * ncp + j_off + j * 3;
        oj[0] = dJ[0].x; oj[1] = dJ[1].x; oj[2] = dJ[2].x;
      }
      if (ta + 1 < a.T) {
        float* o = a.d_rot + ((size_t)(ta + 1) * NB + j) * 9;
#pragma unroll
        for (int e = 0; e < 9; ++e) o[e] = dR[e].y;
        float* oj = a.d_out + (size_t)(ta + 1) * ncp + j_off + j * 3;
        oj[0] = dJ[0].y; oj[1] = dJ[1].y; oj[2] = dJ[2].y;
      }
    }
    CP_STAMP(11)
  }
}

#ifdef EMPOSE_PAIR_TRACE
extern "C" int empose_debug_pair_trace(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pair_trace), sizeof(long long) * 64);
}
#endif

hipError_t launch_chain_pair(const ChainArgs& a, hipStream_t stream) {
  const PairLds L = pair_layout(a.tab.nv, a.tab.ncp, a.tab.max_deg, a.tab.n_chunks);
  const size_t lds = (size_t)L.total * cp::NPAIR * sizeof(float) * 2 + (size_t)a.tab.poff.total * sizeof(uint32_t);
  static size_t attr_set = 0;
  if (lds > attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(chain_pair_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_set = lds;
  }
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipGetLastError();
    n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (size_t)(160 * 1024) / lds));
  const int ntiles = (a.T + 2 * cp::NPAIR - 1) / (2 * cp::NPAIR);
  const int blocks = std::min(ntiles, n_cu * per_cu);
  hipLaunchKernelGGL(chain_pair_kernel, dim3(blocks), dim3(cp::NT), lds, stream, a);
  return hipGetLastError();
}

}  // namespace empose
